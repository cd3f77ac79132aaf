#!/usr/bin/env python3
"""Stress of the host-buffer entries (GPU box): one long camera stream cut into tickets of random sizes (1..max_batch
frames: latency-mode and throughput-mode calls alternate), pageable / pinned / pinned-in-device-layout sources, lone
and pipelined tickets, view and copy-out collects -- every frame's keypoints, descriptors and match table must equal what
the same stream gives one frame per call.  (The one-frame-per-call path is held to the oracle by tests/; this tool
covers orders of magnitude more schedules than a test can afford to check against the CPU.)
usage: stress_host_path.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, synth  # noqa: E402
from orbslamm_amd.extractor import PinnedFrames  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H, NF, MAXB, NFR = 640, 480, 1000, 16, 96
frames = synth.make_frames(W, H, NFR, stream=3)

# expected: the stream one frame per call
ex1 = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1)
ref = []
for f in range(NFR):
    kps, desc, n, m, nm = ex1.extract_match_host(frames[f][None], copy=True)
    ref.append((kps[0][:n[0]].tobytes(), desc[0][:n[0]].tobytes(), m[0][:n[0]].tobytes(), int(nm[0])))
ex1.close()

rng = np.random.default_rng(seed)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=MAXB)
pins = [ex.alloc_pinned_frames(MAXB, W, H) for _ in range(3)]
t_end = time.time() + seconds
tickets, checked, passes = [], 0, 0


def check(res, f0, b, view):
    global checked
    kps, desc, n, m, nm = res
    for f in range(b):
        k = int(n[f])
        exp = ref[f0 + f]
        assert kps[f][:k].tobytes() == exp[0], "keypoints of frame %d" % (f0 + f)
        assert desc[f][:k].tobytes() == exp[1], "descriptors of frame %d" % (f0 + f)
        if f0 + f > 0:
            assert m[f][:k].tobytes() == exp[2] and int(nm[f]) == exp[3], "matches of frame %d" % (f0 + f)
        checked += 1


while time.time() < t_end:
    ex.reset_stream()
    o, i = 0, 0
    depth = int(rng.integers(1, 4))      # 1 = every ticket alone, 3 = the full pipeline
    while o < NFR:
        b = int(min(rng.integers(1, MAXB + 1), NFR - o))
        chunk = frames[o:o + b]
        kind = int(rng.integers(0, 3))
        if kind == 0:
            src = np.ascontiguousarray(chunk)
        else:
            p = pins[i % 3]
            p.array[:b, :, :W] = chunk
            src = [p.array[f, :, :W] for f in range(b)] if kind == 1 else PinnedFrames(p.owner, p.ptr, b, W, H, p.stride, p.pitch)
        tickets.append((ex.submit_host(src), o, b))
        o += b
        i += 1
        while len(tickets) >= depth:
            t, f0, bb = tickets.pop(0)
            check(ex.collect_host(t, view=False), f0, bb, False)
    while tickets:
        t, f0, bb = tickets.pop(0)
        check(ex.collect_host(t, view=False), f0, bb, False)
    passes += 1
print("stress ok: %d passes over a %d-frame stream, %d frames checked, seed %d" % (passes, NFR, checked, seed))
