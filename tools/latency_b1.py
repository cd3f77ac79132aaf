#!/usr/bin/env python3
"""single-frame (drop-in per-frame mode) latency breakdown"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orbslamm_amd import ORBextractor, synth
W, H = 1241, 376
fr = synth.make_frames(W, H, 4)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
for _ in range(5): ex(fr[0])
for serial in (False, True):
    ex.set_serial(serial)
    ex(fr[0])
    t = time.perf_counter(); n = 50
    for i in range(n): ex(fr[i % 4])
    dt = (time.perf_counter() - t) / n
    ex.profile_enable(True); ex.profile_read(True)
    for i in range(n): ex(fr[i % 4])
    p = ex.profile_read(True); ex.profile_enable(False)
    print("serial=%d host call %.1f us; gpu spans (us):" % (serial, dt * 1e6), {k: round(v[0] / n * 1e3, 1) for k, v in p.items() if v[1]})
