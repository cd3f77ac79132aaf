#!/usr/bin/env python3
"""Phase times of one k_proj_fused workgroup (library built with -DORBT_PHASE_TIMING, see tools/proj_phases.sh)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth, _lib  # noqa: E402

W, H, NF = 1241, 376, 2000
fr = synth.make_frames(W, H, 2, stream=0)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, device=0)
m = ORBmatcher(0.9, True, device=0)
sf = np.array(ex.GetScaleFactors(), np.float32)
g = make_grid(0.0, 0.0, float(W), float(H))
fs = m.frame_set(2, ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0] * 5, g, [0.0, float(W), 0.0, float(H)], sf)
ex.extract_batch_device(*ex.upload_frames(fr))
fs.build_from_extractor(0, ex)
L = _lib.lib()
names = ["init+stage lists", "rounds", "write back"]
for th in (15.0, 30.0):
    acc = np.zeros(3)
    for rep in range(20):
        fs.track([1], [0], th=th)
        a, n = fs.results()
        us = (C.c_double * 3)()
        L.orbm_debug_phase_times(us, 3)
        if rep >= 5:
            acc += np.array(list(us))
    acc /= 15
    us2 = (C.c_double * 11)()
    L.orbm_debug_phase_times(us2, 11)
    cn = ["stage grid", "count walk", "scans", "list walk", "hamming", "compact count+alloc", "write"]
    print("  candidates workgroup (0,0): " + ", ".join("%s %.1f" % (k, v) for k, v in zip(cn, list(us2)[4:11])) + " us")
    r, cands = fs.stats(0)
    print("th=%g: %d matches, %d rounds, %d candidates | " % (th, n[0], r, cands) + ", ".join("%s %.1f us" % (k, v) for k, v in zip(names, acc)) + " | total %.1f us" % acc.sum())
