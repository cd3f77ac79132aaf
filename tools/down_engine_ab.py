#!/usr/bin/env python3
"""The pipelined host entries (orbx_submit_batch / orbx_collect_view, three tickets of 64 frames in flight) under the
ORBX_DOWN_ENGINE switch of the library: 0 = results written to the pinned block by k_pack_host's shader stores, 1 (default
for a batch submitted behind others) = gathered into the slot's block in HBM and taken down by ONE engine copy.
Prints frames/s from pinned and from pageable frames and a checksum over every collected ticket's results (equal across
the switch).  tools/down_engine_ab.sh runs the rounds."""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, synth  # noqa: E402

W, H, B = 1241, 376, 64
SECONDS = float(os.environ.get("AB_SECONDS", "1.5"))
DEPTH = int(os.environ.get("AB_DEPTH", "3"))  # tickets in flight (the library holds three slots unless built with -DORBX_HOST_SLOTS=n)
frames = synth.make_frames(W, H, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
pin = [ex.alloc_pinned_frames(B, W, H) for _ in range(3)]
for p in pin:
    p.fill(frames)
out = {}
for name, src in (("pinned", lambda i: pin[i % 3]), ("pageable", lambda i: frames)):
    tick, n, crc = [], 0, 0

    def take(t):
        global crc
        if n_checked[0] < 6:  # the first tickets: every byte of the results (copy-out collect)
            n_checked[0] += 1
            kps, desc, nn, m, nm = ex.collect_host(t, view=False)
            for f in range(B):
                k = int(nn[f])
                crc = zlib.crc32(kps[f, :k].tobytes(), crc)
                crc = zlib.crc32(desc[f, :k].tobytes(), crc)
                crc = zlib.crc32(m[f, :k].tobytes(), crc)
            crc = zlib.crc32(nm.tobytes(), crc)
            ref[0] = (nn.copy(), nm.copy())
        else:  # then what a consumer of the pinned block costs; every ticket repeats the same frames: same counts
            nn, nm = ex.collect_host(t, view=True)
            if not (np.array_equal(nn, ref[0][0]) and np.array_equal(nm, ref[0][1])):
                bad[0] += 1
    n_checked, ref, bad = [0], [None], [0]
    for _ in range(6):  # warm
        tick.append(ex.submit_host(src(n))); n += 1
        if len(tick) == DEPTH:
            take(tick.pop(0))
    while tick:
        take(tick.pop(0))
    n = 0
    t_sub = t_col = 0.0
    t = time.perf_counter()
    while time.perf_counter() - t < SECONDS:
        a = time.perf_counter()
        tick.append(ex.submit_host(src(n)))
        b = time.perf_counter()
        if len(tick) == DEPTH:
            take(tick.pop(0))
        t_sub += b - a
        t_col += time.perf_counter() - b
        n += 1
    while tick:
        take(tick.pop(0))
    dt = time.perf_counter() - t
    out[name] = (n * B / dt, crc, bad[0], t_sub / n * 1e3, t_col / n * 1e3)
# results into caller-owned pinned arrays (orbx_submit_batch_into / orbx_collect), pageable frames, like bench.py's pipelined_into
outs = [ex.alloc_pinned_results(B) for _ in range(3)]
tick, n = [], 0
for _ in range(6):
    tick.append(ex.submit_host_into(frames, outs[n % 3])); n += 1
    if len(tick) == DEPTH:
        ex.collect_into(tick.pop(0))
while tick:
    ex.collect_into(tick.pop(0))
n = 0
t = time.perf_counter()
while time.perf_counter() - t < SECONDS:
    tick.append(ex.submit_host_into(frames, outs[n % 3])); n += 1
    if len(tick) == DEPTH:
        ex.collect_into(tick.pop(0))
while tick:
    ex.collect_into(tick.pop(0))
into_fps = n * B / (time.perf_counter() - t)
crc_into = 0
for f in (0, B // 2, B - 1):
    k = int(outs[0][2][f])
    crc_into = zlib.crc32(outs[0][0][f, :k].tobytes() + outs[0][1][f, :k].tobytes() + outs[0][3][f, :k].tobytes(), crc_into)
print("   into caller arrays (pageable frames): %.0f frames/s  crc %08x" % (into_fps, crc_into))
print("ORBX_DOWN_ENGINE=%s  pinned %.0f frames/s  pageable %.0f frames/s  crc of the first six tickets %08x %08x  tickets with other counts %d"
      % (os.environ.get("ORBX_DOWN_ENGINE", "(default)"), out["pinned"][0], out["pageable"][0], out["pinned"][1], out["pageable"][1],
         out["pinned"][2] + out["pageable"][2]))
print("   host side per ticket: pinned submit %.3f ms + collect %.3f ms; pageable submit %.3f + collect %.3f" % (out["pinned"][3], out["pinned"][4], out["pageable"][3], out["pageable"][4]))
