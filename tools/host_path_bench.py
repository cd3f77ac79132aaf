#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (orbx_extract_batch: pageable host frames in,
keypoints/descriptors back on the host).  Reported in DESIGN.md, never as bench `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, synth  # noqa: E402

W, H, B = 1241, 376, 64
frames = synth.make_frames(W, H, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
for b in (1, 8, 64):
    ex.extract_batch(frames[:b])
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        ex.extract_batch(frames[:b])
    dt = (time.perf_counter() - t) / n
    print("host path B=%d: %.3f ms per call, %.0f frames/s (incl. staging copy, H2D, D2H, numpy slicing)" % (b, dt * 1e3, b / dt))
