#!/usr/bin/env python3
"""The live-stream loop as the reference runs it (one frame per robot per iteration: M/Examples/Monocular/mono_kitti.cc:80-101,
Tracking.cc:240-267, 925-936), through the Python mirror of the C ABI.
  live_loop.py track N        one stream: extract + Frame tail + SearchByProjection(Cur, Last), results on the host, per call
  live_loop.py track2 N       the same with the extraction of frame t+1 submitted before the search of frame t is collected
  live_loop.py bf N           one stream: extract + brute-force match vs previous frame
  live_loop.py threads K N    K handles on one GPU, one Python thread each (mode track)
Prints median / mean per frame like mono_tum.cc:113-122.  (Python threads share the GIL: examples/multi_robot is the
native form of the K-stream loop.)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth  # noqa: E402

W, H, NF = 1241, 376, 2000


class Stream:
    def __init__(self, stream=0, device=0):
        self.fr = synth.make_frames(W, H, 8, stream=stream)
        self.ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=1, device=device)
        self.m = ORBmatcher(0.9, True, device=device)
        sf = np.array(self.ex.GetScaleFactors(), np.float32)
        g = make_grid(0.0, 0.0, float(W), float(H))
        self.fs = self.m.frame_set(4, self.ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0, 0, 0, 0, 0], g, [0.0, float(W), 0.0, float(H)], sf)
        if os.environ.get("LIVE_ATTACH", "1") == "1":
            self.fs.attach(self.ex)

    def live(self, n):
        """the attached chain: frame + build + search submitted together, keypoints and the match table collected after"""
        ex, fs, fr = self.ex, self.fs, self.fr
        lat = []
        for i in range(n):
            slot, prev = i & 1, (i & 1) ^ 1
            t0 = time.perf_counter()
            tk = ex.submit_host(fr[i % 8][None], match=False)
            fs.build_from_extractor(slot, ex)
            if i:
                fs.track([slot], [prev], th=15.0)
            ex.collect_host(tk, view=True)
            if i:
                fs.results()
            lat.append(time.perf_counter() - t0)
        return np.array(lat[5:]) * 1e3

    def track(self, n):
        ex, fs, fr = self.ex, self.fs, self.fr
        lat = []
        for i in range(n):
            slot, prev = i & 1, (i & 1) ^ 1
            t0 = time.perf_counter()
            ex.extract_match_host(fr[i % 8][None], match=False)
            fs.build_from_extractor(slot, ex)
            if i:
                fs.track([slot], [prev], th=15.0)
                fs.results()
            else:
                fs.sync()
            lat.append(time.perf_counter() - t0)
        return np.array(lat[5:]) * 1e3

    def track2(self, n):
        """depth 2: frame t+1 is uploaded and extracted while the search of frame t runs and is collected"""
        ex, fs, fr = self.ex, self.fs, self.fr
        lat = []
        tk = ex.submit_host(fr[0][None], match=False)
        fs.build_from_extractor(0, ex)
        t0 = time.perf_counter()
        for i in range(1, n):
            slot = i & 3
            tk2 = ex.submit_host(fr[i % 8][None], match=False)
            fs.build_from_extractor(slot, ex)
            ex.collect_host(tk, view=True)          # keypoints + descriptors of frame i-1 on the host
            if i > 1:
                fs.track([(i - 1) & 3], [(i - 2) & 3], th=15.0)  # the search of frame i-1 (its pose would come from frame i-2's result)
                fs.results()
            tk = tk2
            t1 = time.perf_counter()
            lat.append(t1 - t0)
            t0 = t1
        ex.collect_host(tk, view=True)
        return np.array(lat[5:]) * 1e3

    def bf(self, n):
        ex, fr = self.ex, self.fr
        lat = []
        for i in range(n):
            t0 = time.perf_counter()
            ex.extract_match_host(fr[i % 8][None])
            lat.append(time.perf_counter() - t0)
        return np.array(lat[5:]) * 1e3


def report(name, lat):
    print("%-10s median %.4f ms  mean %.4f ms  p95 %.4f  -> %.0f frames/s (%d frames)" % (name, np.median(lat), lat.mean(), np.percentile(lat, 95), 1e3 / lat.mean(), len(lat)))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "track"
    if mode == "threads":
        K, n = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 400
        ss = [Stream(k) for k in range(K)]
        for s in ss:
            s.track(20)
        out = [None] * K
        th = [threading.Thread(target=lambda k=k: out.__setitem__(k, ss[k].track(n))) for k in range(K)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        allv = np.concatenate(out)
        report("K=%d" % K, allv)
        print("aggregate %.0f frames/s" % (K * n / dt))
        return
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    s = Stream()
    getattr(s, mode)(20)
    report(mode, getattr(s, mode)(n))


if __name__ == "__main__":
    main()
