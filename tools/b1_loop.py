#!/usr/bin/env python3
"""N calls of the one-frame-per-call entry (run under rocprofv3 --kernel-trace by tools/b1_timeline.sh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, synth  # noqa: E402

W, H = 1241, 376
fr = synth.make_frames(W, H, 8)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    ex.extract_match_host(fr[i % 8][None])
