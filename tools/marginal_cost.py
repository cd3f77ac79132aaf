#!/usr/bin/env python3
"""On the GPU box: what the matching costs the overlapped pipeline -- frames/s of extract + match against extract
alone, same frames, same handle, 200 steps each, alternating (usage: python tools/marginal_cost.py [rounds]).
Alone on the GPU the matching kernels take 0.115 ms per 64-frame step (19 % of the sum of the isolated kernel times);
this measures their share of the STEP when they run beside the next batch's extraction."""
import sys
import time

sys.path.insert(0, ".")
from orbslamm_amd import ORBextractor, synth

W, H, B, K = 1241, 376, 64, 200
fr = synth.make_frames(W, H, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B, device=0)
d = ex.upload_frames(fr, stride=1280)


def run(with_match):
    for _ in range(5):
        ex.extract_batch_device(*d)
        if with_match:
            ex.match_prev_batch_device(0.7, 50, True)
    ex.sync()
    t0 = time.perf_counter()
    for _ in range(K):
        ex.extract_batch_device(*d)
        if with_match:
            ex.match_prev_batch_device(0.7, 50, True)
    ex.sync()
    dt = time.perf_counter() - t0
    return B * K / dt, dt / K * 1e3


for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    a = run(True)
    b = run(False)
    print("extract+match %7.0f frames/s %.4f ms/step | extract only %7.0f frames/s %.4f ms/step | matching costs %.4f ms/step (%.1f %%)"
          % (a[0], a[1], b[0], b[1], a[1] - b[1], 100 * (a[1] - b[1]) / a[1]))
