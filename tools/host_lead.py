#!/usr/bin/env python3
"""How far ahead of the GPU does the host run in bench.py's loop?  Prints the time the host needs to ENQUEUE
K steps and the time until the GPU has finished them, with and without the in-kernel-stream HIP-event profiling."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from orbslamm_amd import ORBextractor, synth
B, K = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 200
fr = synth.make_frames(1241, 376, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=B, device=0)
d = ex.upload_frames(fr, stride=1280)
def step():
    ex.extract_batch_device(*d); ex.match_prev_batch_device(0.7, 50, True)
for prof in (False, True, False, True):
    for _ in range(5): step()
    ex.sync()
    ex.profile_enable(prof); ex.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    ex.sync()
    t2 = time.perf_counter()
    ex.profile_read(reset=True); ex.profile_enable(False)
    print("profile=%d: host enqueue %.1f us/step, GPU done %.1f us/step  (%d steps, %.0f frames/s)" % (prof, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, K, B * K / (t2 - t0)))
