#!/usr/bin/env python3
"""Differential soak of the frame-set searches with caller-supplied queries and of the BoW side (test infrastructure, like
tests/; companion of tests/soak/fuzz_frontend.py):
  * Tracking::SearchLocalPoints' search (orbm_track_local_points, mode 3): a random local map against a resident frame,
    th in {1, 3, 5}, with and without features already taken;
  * TrackWithMotionModel's search with a caller's pose (orbm_track_frame_projected, modes 4 and 5): a random similarity
    applied to LastFrame's keypoints, random validity / observation flags;
  * Frame::ComputeBoW for random slot ranges + SearchByBoW(KeyFrame, Frame) over random slot pairs on random trees.
On the GPU box:  python tests/soak/fuzz_tracking.py [rounds] [seed] > gpurun_out/fuzz_tracking.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth  # noqa: E402
from vocab_cases import make_vocab  # noqa: E402


def local_map_queries(rng, frames, sf, th, nq_target):
    """tests/test_gpu_tracking.py: a local map's worth of projected MapPoints made from neighbouring frames' keypoints"""
    ks = np.concatenate([k for k, _ in frames]); ds = np.concatenate([d for _, d in frames])
    sel = rng.permutation(len(ks))[:nq_target]
    ks, ds = ks[sel], ds[sel]
    nq = len(ks)
    lvl = ks["octave"].astype(np.int32)
    r = (np.where(rng.random(nq) < 0.5, np.float32(2.5), np.float32(4.0)) * np.float32(th)).astype(np.float32) * sf[lvl]
    uvr = np.stack([ks["x"] + rng.normal(0, 1.2, nq), ks["y"] + rng.normal(0, 1.2, nq), r], axis=1).astype(np.float32)
    ql = np.stack([lvl - 1, lvl], axis=1).astype(np.int8)
    qv = (rng.random(nq) < 0.95).astype(np.uint8)
    qo = (rng.random(nq) < 0.9).astype(np.uint8)
    return uvr, ql, np.ascontiguousarray(ds), qv, qo


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    nfmax = int(sys.argv[3]) if len(sys.argv) > 3 else 2400   # features per frame below this (above ~4000 the set's BoW sort leaves LDS;
    rng = np.random.default_rng(seed)                         # above 8192 every search and the frame build take their memory forms)
    shapes = [(640, 480), (1241, 376), (752, 480), (512, 384)]
    if nfmax > 5000:
        shapes = [(1920, 1080), (1400, 1000), (1241, 376)]    # frames that can hold that many features
    counts = {"local_points": 0, "projected": 0, "bow_vectors": 0, "bow_pairs": 0}
    t0 = time.time()
    for r in range(rounds):
        w, h = shapes[int(rng.integers(0, len(shapes)))]
        nf = int(rng.integers(300, nfmax))
        B = int(rng.integers(3, 7))
        fr = synth.make_frames(w, h, B, stream=int(rng.integers(0, 500)))
        gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
        sf = np.array(gex.GetScaleFactors(), np.float32)
        bounds = [0.0, float(w), 0.0, float(h)]
        g = make_grid(0.0, 0.0, float(w), float(h))
        gp = ob.make_grid_params(0.0, 0.0, float(w), float(h))
        ratio = float(np.float32(rng.choice([0.7, 0.8, 0.9])))
        m = ORBmatcher(ratio, True, device=0)
        fs = m.frame_set(B + 1, gex.max_keypoints, [0.8 * w, 0.8 * w, w / 2.0, h / 2.0], [0, 0, 0, 0, 0], g, bounds, sf)
        gex.extract_batch_device(*gex.upload_frames(fr))
        fs.build_from_extractor(1, gex)   # frames 0..B-1 -> slots 1..B
        host = [gex.download(f) for f in range(B)]
        # ---- SearchLocalPoints' search
        for _ in range(3):
            t = int(rng.integers(0, B))
            kc, dc = host[t]
            start, idx = ob.grid_build(gp, kc)
            others = [host[i] for i in range(B) if i != t]
            th = float(rng.choice([1.0, 3.0, 5.0])); nq = int(rng.integers(1, max(5000, 2 * nf)))
            uvr, ql, qd, qv, qo = local_map_queries(rng, others, sf, th, nq)
            with_occ = bool(rng.integers(0, 2))
            occ = (rng.random(len(kc)) < 0.3).astype(np.uint8) if with_occ else np.zeros(len(kc), np.uint8)
            fs.track_local_points(1 + t, uvr, ql, qd, qv, qo, occ if with_occ else None, nnratio=ratio)
            assign, nm = fs.results()
            wa, _, wn = ob.search_by_projection(3, ratio, True, 100, uvr, ql, qd, None, qv, qo, gp, kc, start, idx, dc, occ, np.full(len(kc), -1, np.int32))
            if nm[0] != wn or not np.array_equal(assign[0, :len(kc)], wa):
                print("DIFFERENCE track_local_points", dict(round=r, w=w, h=h, nf=nf, th=th, nq=len(uvr), occ=with_occ, ratio=ratio, seed=seed))
                return 1
            counts["local_points"] += 1
        # ---- the search with a caller's pose
        for _ in range(3):
            c, l = int(rng.integers(0, B)), int(rng.integers(0, B))
            (kl, dl), (kc, dc) = host[l], host[c]
            start, idx = ob.grid_build(gp, kc)
            mode = int(rng.choice([4, 4, 5])); th = float(rng.choice([7.0, 10.0, 15.0, 30.0])); ori = bool(rng.integers(0, 2)); thd = int(rng.choice([64, 100]))
            a = np.deg2rad(float(rng.uniform(-1.5, 1.5))); sc = float(rng.uniform(0.99, 1.01)); tx, ty = float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4))
            cx, cy = w / 2.0, h / 2.0
            u = (np.cos(a) * (kl["x"] - cx) - np.sin(a) * (kl["y"] - cy)) * sc + cx + tx
            v = (np.sin(a) * (kl["x"] - cx) + np.cos(a) * (kl["y"] - cy)) * sc + cy + ty
            uvr = np.stack([u, v, np.float32(th) * sf[kl["octave"]]], axis=1).astype(np.float32)
            lvl = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
            qv = ((rng.random(len(kl)) < 0.85) & (u >= 0) & (u <= w) & (v >= 0) & (v <= h)).astype(np.uint8)
            qo = (rng.random(len(kl)) < 0.9).astype(np.uint8)
            with_occ = bool(rng.integers(0, 2))
            occ = (rng.random(len(kc)) < 0.2).astype(np.uint8) if with_occ else np.zeros(len(kc), np.uint8)
            fs.track_projected(1 + c, 1 + l, uvr, lvl, qv, qo, occ if with_occ else None, th_dist=thd, nnratio=ratio, check_ori=ori, mode=mode)
            assign, nm = fs.results()
            wa, _, wn = ob.search_by_projection(mode, ratio, ori, thd, uvr, lvl, dl, kl["angle"], qv, qo, gp, kc, start, idx, dc, occ, np.full(len(kc), -1, np.int32))
            if nm[0] != wn or not np.array_equal(assign[0, :len(kc)], wa):
                print("DIFFERENCE track_projected", dict(round=r, w=w, h=h, nf=nf, mode=mode, th=th, thd=thd, ori=ori, pair=(c, l), ratio=ratio, seed=seed))
                return 1
            counts["projected"] += 1
        # ---- ComputeBoW + SearchByBoW on the set
        k, L = int(rng.integers(3, 11)), int(rng.integers(2, 5))
        voc = make_vocab(rng, k, L, ragged=bool(rng.integers(0, 2)))
        levelsup = int(rng.integers(0, L + 2))
        G = ORBVocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
        O = ob.Vocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        fs.compute_bow(G, 1, B, levelsup)
        fvs = []
        for f in range(B):
            keys, desc = host[f]
            (owid, owval), ofv = O.transform(desc, levelsup)
            wid, wval = fs.bow_vector(1 + f)
            if not np.array_equal(wid, owid) or wval.tobytes() != owval.tobytes():
                print("DIFFERENCE ComputeBoW", dict(round=r, k=k, L=L, levelsup=levelsup, frame=f, seed=seed))
                return 1
            fvs.append(ofv)
            counts["bow_vectors"] += 1
        npairs = int(rng.integers(1, B + 1))
        kf = rng.integers(1, B + 1, npairs); cur = rng.integers(1, B + 1, npairs)
        ori = bool(rng.integers(0, 2))
        fs.search_by_bow(kf, cur, nnratio=ratio, check_ori=ori)
        match, nm = fs.bow_results()
        for p in range(npairs):
            (kq, dq), (kt, dt) = host[kf[p] - 1], host[cur[p] - 1]
            want, wn = ob.search_by_bow(dq, kq["angle"], None, fvs[kf[p] - 1], dt, kt["angle"], None, fvs[cur[p] - 1], ratio, ori, True)
            if nm[p] != wn or not np.array_equal(match[p, :len(kt)], want):
                print("DIFFERENCE SearchByBoW on the set", dict(round=r, k=k, L=L, levelsup=levelsup, pair=(int(kf[p]), int(cur[p])), ratio=ratio, ori=ori, seed=seed))
                return 1
            counts["bow_pairs"] += 1
        fs.close(); m.close(); G.close(); gex.close()
    print("tracking soak: %d rounds (seed %d): %s -- every match table, BowVector (ids and binary64 values) equal to the oracle's; %.0f s"
          % (rounds, seed, counts, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
