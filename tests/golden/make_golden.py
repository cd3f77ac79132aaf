#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

PROVENANCE: these vectors are produced by the CPU ORACLE (oracle/), not by the
reference binary: the reference's ORBextractor cannot be built here (OpenCV is
neither vendored nor installed) and the reference ships no test
vectors (F4).  They pin the oracle against regressions and give the GPU tests a
committed target; they do NOT pin the oracle against OpenCV ("parity unpinned").
Inputs are regenerated from seeds (orbslamm_amd/synth.py), only outputs are stored.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import synth  # noqa: E402

CASES = [  # name, w, h, nfeatures, scale, levels, ini, min, stream
    ("small_320x240_500", 320, 240, 500, 1.2, 8, 20, 7, 3),
    ("odd_401x263_700", 401, 263, 700, 1.2, 8, 20, 7, 4),
    ("tum_640x480_1000", 640, 480, 1000, 1.2, 8, 20, 7, 0),
    ("kitti_1241x376_2000", 1241, 376, 2000, 1.2, 8, 20, 7, 0),
]


def main():
    for name, w, h, nf, sf, nl, ini, mn, stream in CASES:
        fr = synth.make_frames(w, h, 2, stream=stream)
        ex = ob.Extractor(nf, sf, nl, ini, mn)
        r0, r1 = ex(fr[0]), ex(fr[1])
        m, n = ob.match_bruteforce(r1["desc"], r1["kps"]["angle"], r0["desc"], r0["kps"]["angle"], 0.7, 50, True)
        big = w * h > 400 * 300
        out = dict(params=np.array([w, h, nf, nl, ini, mn, stream], dtype=np.int64), scale=np.float32(sf),
                   frame_sha=np.frombuffer(hashlib.sha256(fr.tobytes()).digest(), dtype=np.uint8),
                   nmatch=np.int64(n), cand_counts=np.stack([r0["cand_counts"], r1["cand_counts"]]),
                   kept_counts=np.stack([r0["kept_counts"], r1["kept_counts"]]))
        for i, r in enumerate((r0, r1)):
            if big:  # large cases: digests only
                out["kps%d_sha" % i] = np.frombuffer(hashlib.sha256(r["kps"].tobytes()).digest(), dtype=np.uint8)
                out["desc%d_sha" % i] = np.frombuffer(hashlib.sha256(r["desc"].tobytes()).digest(), dtype=np.uint8)
            else:
                out["kps%d" % i] = r["kps"]
                out["desc%d" % i] = r["desc"]
        if big:
            out["match_sha"] = np.frombuffer(hashlib.sha256(m.tobytes()).digest(), dtype=np.uint8)
        else:
            out["match"] = m
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, len(r0["kps"]), len(r1["kps"]), n)


if __name__ == "__main__":
    main()
