#!/usr/bin/env python3
"""The host-buffer entries, PCIe inclusive (never bench `value`; bench.py reports the same figures as `host_path`):
  b1        orbx_extract_match_batch with one pageable frame per call: per-frame latency (median / mean, like
            mono_tum.cc:113-122) with the GPU spans of the stages; also from pinned frames and without matching
  batches   synchronous B = 8 / 64
  pipeline  orbx_submit_batch / orbx_collect_view, three tickets in flight, pageable and pinned frames"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, synth  # noqa: E402

W, H, B = 1241, 376, 64
frames = synth.make_frames(W, H, B)


def lat(ex, src, n=300, match=True):
    for i in range(10):
        ex.extract_match_host(src(i), match=match)
    ts = []
    for i in range(n):
        t = time.perf_counter()
        ex.extract_match_host(src(i), match=match)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts) * 1e3
    return float(np.median(ts)), float(ts.mean()), float(np.percentile(ts, 99))


ex1 = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
pin1 = [ex1.alloc_pinned_frames(1, W, H) for _ in range(4)]
for i, p in enumerate(pin1):
    p.fill(frames[i:i + 1])
for name, src, match in (("pageable, extract+match", lambda i: frames[i % 8][None], True),
                         ("pinned,   extract+match", lambda i: pin1[i % 4], True),
                         ("pageable, extract only ", lambda i: frames[i % 8][None], False)):
    md, mn, p99 = lat(ex1, src, match=match)
    print("B=1 %s: median %.3f ms  mean %.3f ms  p99 %.3f ms  (%.0f frames/s)" % (name, md, mn, p99, 1e3 / mn))
ex1.profile_enable(True)
ex1.profile_read(True)
n = 100
for i in range(n):
    ex1.extract_match_host(frames[i % 8][None])
p = ex1.profile_read(True)
ex1.profile_enable(False)
print("B=1 gpu spans (us):", {k: round(v[0] / n * 1e3, 1) for k, v in p.items() if v[1]})
ex1.close()

ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
for b in (8, 64):
    ex.extract_match_host(frames[:b])
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        ex.extract_match_host(frames[:b])
    dt = (time.perf_counter() - t) / n
    print("synchronous B=%d pageable: %.3f ms per call, %.0f frames/s" % (b, dt * 1e3, b / dt))
pin = [ex.alloc_pinned_frames(B, W, H) for _ in range(3)]
for p in pin:
    p.fill(frames)
for name, src in (("pageable", lambda i: frames), ("pinned", lambda i: pin[i % 3])):
    for view in (True, False):
        tick, n = [], 0
        t = time.perf_counter()
        while time.perf_counter() - t < 1.0:
            tick.append(ex.submit_host(src(n)))
            if len(tick) == 3:
                ex.collect_host(tick.pop(0), view=view)
            n += 1
        while tick:
            ex.collect_host(tick.pop(0), view=view)
        dt = time.perf_counter() - t
        print("pipelined B=64 %s, %s: %.0f frames/s (%.2f GB/s of frames)" % (name, "view collect" if view else "copy-out collect",
                                                                             n * B / dt, n * B * W * H / dt / 1e9))
