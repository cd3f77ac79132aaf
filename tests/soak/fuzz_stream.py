#!/usr/bin/env python3
"""Stateful differential soak of ONE extractor handle as a camera stream (test infrastructure under tests/): a random
schedule of the host entries (synchronous calls, pipelined tickets up to three in flight, collected as copies), the
device-resident entry (upload + extract + match + per-frame downloads), stream resets and SHAPE CHANGES on the same
handle, batch sizes 1..8 so that latency-mode and throughput-mode calls alternate.  Every frame's keypoints, descriptors
and match table against the stream's previous frame are compared with the oracle's.  On the GPU box:
    python tests/soak/fuzz_stream.py [ops] [seed] > gpurun_out/fuzz_stream.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import ORBextractor, synth  # noqa: E402


def main():
    nops = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    rng = np.random.default_rng(seed)
    nf, maxB = 700, 8
    shapes = [(640, 480), (401, 263), (752, 480), (320, 240)]
    pools = {s: synth.make_frames(s[0], s[1], 48, stream=10 + i) for i, s in enumerate(shapes)}
    cursor = {s: 0 for s in shapes}
    ex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=800, max_height=600, max_batch=maxB, device=0)
    oex = ob.Extractor(nf, 1.2, 8, 20, 7)
    cache = {}

    def ref_of(shape, idx):
        key = (shape, idx)
        if key not in cache:
            cache[key] = oex(pools[shape][idx])
        return cache[key]

    prev = None          # oracle record of the stream's previous frame
    inflight = []        # (ticket, [(shape, idx)...], prev-at-submit chain handled at submit time)
    expected = {}        # ticket -> list of (ref, match table, nmatch)
    shape = handle_shape = shapes[0]
    frames_done = 0
    kinds = {"sync": 0, "ticket": 0, "device": 0, "reset": 0, "shape": 0}
    t0 = time.time()

    def take(b):
        idx = [(cursor[shape] + i) % 48 for i in range(b)]
        cursor[shape] = (cursor[shape] + b) % 48
        return idx

    def expect(idx):
        nonlocal prev
        out = []
        for i in idx:
            r = ref_of(shape, i)
            if prev is None:
                m, n = np.full(len(r["kps"]), -1, np.int32), 0
            else:
                m, n = ob.match_bruteforce(r["desc"], r["kps"]["angle"], prev["desc"], prev["kps"]["angle"], 0.7, 50, True)
            out.append((r, np.asarray(m), int(n)))
            prev = r
        return out

    def check(exp, kps, desc, n, m, nm, what):
        r, em, en = exp
        n = int(n)
        if n != len(r["kps"]) or kps[:n].tobytes() != r["kps"].tobytes() or not np.array_equal(desc[:n], r["desc"]):
            print("DIFFERENCE (keypoints/descriptors) in", what, "op", kinds, "seed", seed)
            return False
        if int(nm) != en or not np.array_equal(np.asarray(m)[:n], em):
            print("DIFFERENCE (match table: %d vs %d) in" % (int(nm), en), what, "op", kinds, "seed", seed)
            return False
        return True

    def collect_one():
        nonlocal frames_done
        t, exp = inflight.pop(0)
        kps, desc, n, m, nm = ex.collect_host(t, view=False)
        for f, e in enumerate(exp):
            if not check(e, kps[f], desc[f], n[f], m[f], nm[f], "ticket"):
                return False
        frames_done += len(exp)
        return True

    for op in range(nops):
        p = rng.uniform()
        if p < 0.06:
            while inflight:
                if not collect_one():
                    return 1
            ex.reset_stream()
            prev = None
            kinds["reset"] += 1
            continue
        if p < 0.16:
            while inflight:   # a shape change re-configures the handle: outstanding tickets are collected first
                if not collect_one():
                    return 1
            shape = shapes[int(rng.integers(0, len(shapes)))]
            kinds["shape"] += 1
            continue
        if shape != handle_shape:   # include/orbslamm_hip.h: a CALL with another shape than the handle's current one starts a new stream
            prev = None
            handle_shape = shape
        b = int(rng.integers(1, maxB + 1))
        idx = take(b)
        fr = np.ascontiguousarray(pools[shape][idx])
        if p < 0.45:      # synchronous host call
            while inflight:
                if not collect_one():
                    return 1
            exp = expect(idx)
            kps, desc, n, m, nm = ex.extract_match_host(fr, copy=True)
            for f, e in enumerate(exp):
                if not check(e, kps[f], desc[f], n[f], m[f], nm[f], "sync"):
                    return 1
            frames_done += b
            kinds["sync"] += 1
        elif p < 0.75:    # pipelined ticket
            exp = expect(idx)
            inflight.append((ex.submit_host(fr), exp))
            kinds["ticket"] += 1
            while len(inflight) == 3 or (inflight and rng.integers(0, 3) == 0):
                if not collect_one():
                    return 1
        else:             # device-resident entry
            while inflight:
                if not collect_one():
                    return 1
            exp = expect(idx)
            d = ex.upload_frames(fr)
            ex.extract_batch_device(*d)
            ex.match_prev_batch_device(0.7, 50, True)
            for f, e in enumerate(exp):
                k, dd = ex.download(f)
                mm, nmm = ex.download_matches(f)
                if not check(e, k, dd, len(k), mm, nmm, "device"):
                    return 1
            frames_done += b
            kinds["device"] += 1
    while inflight:
        if not collect_one():
            return 1
    print("stream soak: %d operations on one handle (seed %d): %s; %d frames, every keypoint record, descriptor and match table "
          "(against the stream's previous frame, across shape changes and resets) equal to the oracle's; %.0f s"
          % (nops, seed, kinds, frames_done, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
