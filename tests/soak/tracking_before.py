#!/usr/bin/env python3
"""Round-3 'before' stopwatch of the Tracking-shaped matchers through the ROUND-2 entry points (one frame pair per
call, queries from pageable host memory): orbm_frame_create, orbm_search_by_projection_frame (mode 4, identity pose,
th = 15, octave +-1 -- SURVEY.md 8d), orbm_frame_compute_bow + orbm_search_by_bow_frames on a synthetic k=10 / L=6
vocabulary.  Prints one JSON record; kept as the baseline the new tracking path is compared with."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def full_vocab(k, L, seed=7):
    """complete k-ary tree of perturbed random descriptors, breadth-first (loadFromTextFile order), vectorised"""
    rng = np.random.default_rng(seed)
    parent, desc, is_leaf = [], [], []
    prev_ids = np.array([0])
    prev_desc = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    next_id = 1
    for lvl in range(1, L + 1):
        n = len(prev_ids) * k
        d = np.repeat(prev_desc, k, axis=0)
        nflip = max(4, 60 >> (lvl - 1))
        bits = rng.integers(0, 256, (n, nflip))
        for j in range(nflip):
            np.bitwise_xor.at(d, (np.arange(n), bits[:, j] >> 3), (1 << (bits[:, j] & 7)).astype(np.uint8))
        parent.append(np.repeat(prev_ids, k))
        desc.append(d)
        is_leaf.append(np.full(n, lvl == L, np.uint8))
        prev_ids = np.arange(next_id, next_id + n)
        prev_desc = d
        next_id += n
    w = rng.uniform(0.1, 9.0, next_id - 1)
    return dict(parent=np.concatenate(parent).astype(np.int32), is_leaf=np.concatenate(is_leaf), desc=np.concatenate(desc),
                weight=w, k=k, L=L)


def main():
    from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth
    from oracle import binding as ob
    W, H, NF = 1241, 376, 2000
    N = int(os.environ.get("TRK_FRAMES", "60"))
    fr = synth.make_frames(W, H, N + 1, stream=0)
    ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, device=0)
    m = ORBmatcher(0.9, True, device=0)
    voc = full_vocab(10, 6)
    G = ORBVocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
    g = make_grid(0.0, 0.0, float(W), float(H))
    K, D0 = [718.856, 718.856, 607.1928, 185.2157], [0, 0, 0, 0, 0]
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    t = {k: [] for k in ("extract", "frame", "proj", "bow_transform", "bow_search", "total")}
    last = None
    nm_proj, nm_bow = [], []
    for i in range(N + 1):
        t0 = time.perf_counter()
        kps, desc = ex(fr[i])
        t1 = time.perf_counter()
        dk, dd, _, cap = ex.device_results()
        F = m.frame_from_device(dk, dd, len(kps), K, D0, g)  # frame 0 of the one-frame batch
        t2 = time.perf_counter()
        if last is not None:
            lk, ld, LF = last
            nq = len(lk)
            uvr = np.stack([lk["x"], lk["y"], 15.0 * sf[lk["octave"]]], axis=1).astype(np.float32)
            lvl = np.stack([lk["octave"] - 1, lk["octave"] + 1], axis=1).astype(np.int8)
            t3 = time.perf_counter()
            a, occ, n = m.SearchByProjectionFrame(4, 100, uvr, lvl, ld, lk["angle"], None, None, F, np.zeros(len(kps), np.uint8),
                                                  np.full(len(kps), -1, np.int32))
            t4 = time.perf_counter()
            m.frame_compute_bow(F, G, 4)
            t5 = time.perf_counter()
            m.mfNNratio = 0.7
            mm, nb = m.SearchByBoWFrames(LF, None, F, None, True)
            m.mfNNratio = 0.9
            t6 = time.perf_counter()
            nm_proj.append(n); nm_bow.append(nb)
            if i > 5:
                t["extract"].append(t1 - t0); t["frame"].append(t2 - t1); t["proj"].append(t4 - t3)
                t["bow_transform"].append(t5 - t4); t["bow_search"].append(t6 - t5); t["total"].append(t6 - t0)
            m.frame_destroy(LF)
        else:
            m.frame_compute_bow(F, G, 4)
        last = (kps, desc, F)
    out = {"what": "round-2 entry points, one frame pair per call, 1241x376 / 2000 features", "frames": len(t["total"]),
           "matches_proj_mean": float(np.mean(nm_proj)), "matches_bow_mean": float(np.mean(nm_bow))}
    for k, v in t.items():
        v = np.array(v) * 1e3
        out[k + "_ms_median"] = float(np.median(v)); out[k + "_ms_mean"] = float(v.mean())
    # the CPU oracle on the same pair
    O = ob.Vocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    lk, ld, _ = last
    gp = ob.make_grid_params(0.0, 0.0, float(W), float(H))
    t0 = time.perf_counter()
    for _ in range(5):
        start, idx = ob.grid_build(gp, lk)
    t1 = time.perf_counter()
    uvr = np.stack([lk["x"], lk["y"], 15.0 * sf[lk["octave"]]], axis=1).astype(np.float32)
    lvl = np.stack([lk["octave"] - 1, lk["octave"] + 1], axis=1).astype(np.int8)
    for _ in range(5):
        ob.search_by_projection(4, 0.9, True, 100, uvr, lvl, ld, lk["angle"], None, None, gp, lk, start, idx, ld,
                                np.zeros(len(lk), np.uint8), np.full(len(lk), -1, np.int32))
    t2 = time.perf_counter()
    for _ in range(3):
        bv, fv = O.transform(ld, 4)
    t3 = time.perf_counter()
    for _ in range(5):
        ob.search_by_bow(ld, lk["angle"], None, fv, ld, lk["angle"], None, fv, 0.7, True, True)
    t4 = time.perf_counter()
    out["cpu_oracle_ms"] = {"grid": (t1 - t0) / 5 * 1e3, "proj": (t2 - t1) / 5 * 1e3, "bow_transform": (t3 - t2) / 3 * 1e3,
                            "bow_search": (t4 - t3) / 5 * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
