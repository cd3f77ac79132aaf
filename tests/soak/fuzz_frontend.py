#!/usr/bin/env python3
"""Differential soak of the Tracking-shaped front-end against the CPU oracle (test infrastructure under tests/):
  * frame sets: extractor -> UndistortKeyPoints + grid on the device -> batched SearchByProjection(Cur, Last) over random
    pairs of slots, random camera / distortion / bounds / thresholds / batch sizes / shapes;
  * Frame::ComputeStereoMatches on rectified pairs with random band disparities, baselines and shapes;
  * ORBVocabulary::transform on random trees (branching, depth, raggedness, scoring, weighting, levelsup).
On the GPU box:  python tests/soak/fuzz_frontend.py [rounds] [seed] > gpurun_out/fuzz_frontend.txt"""
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


def identity_queries(keys_un, sf, th, bounds):
    uvr = np.stack([keys_un["x"], keys_un["y"], (np.float32(th) * sf[keys_un["octave"]]).astype(np.float32)], axis=1).astype(np.float32)
    lvl = np.stack([keys_un["octave"] - 1, keys_un["octave"] + 1], axis=1).astype(np.int8)
    x, y = keys_un["x"], keys_un["y"]
    qv = ~((x < bounds[0]) | (x > bounds[1]) | (y < bounds[2]) | (y > bounds[3]))
    return uvr, lvl, qv.astype(np.uint8)


def levels_of(oex, pyr, w, h, nl):
    out, o = [], 0
    for l in range(nl):
        lw, lh = oex.level_size(w, h, l)
        out.append(pyr[o:o + lw * lh].reshape(lh, lw))
        o += lw * lh
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nfmax = int(sys.argv[3]) if len(sys.argv) > 3 else 2400   # features per frame below this (above ~4000 the set's BoW sort leaves LDS)
    rng = np.random.default_rng(seed)
    shapes = [(640, 480), (1241, 376), (752, 480), (401, 263), (512, 384)]
    counts = {"frames_built": 0, "track_pairs": 0, "stereo_pairs": 0, "transforms": 0}
    t0 = time.time()
    for r in range(rounds):
        # ------------------------------------------------------------ frame set + batched tracking search
        w, h = shapes[int(rng.integers(0, len(shapes)))]
        nf = int(rng.integers(200, nfmax))
        B = int(rng.integers(2, 9))
        fr = synth.make_frames(w, h, B, stream=int(rng.integers(0, 500)))
        gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
        sf = np.array(gex.GetScaleFactors(), np.float32)
        gex.extract_batch_device(*gex.upload_frames(fr))
        fx = float(rng.uniform(0.6, 1.3)) * w
        K = [fx, fx * float(rng.uniform(0.98, 1.02)), w / 2 + float(rng.uniform(-20, 20)), h / 2 + float(rng.uniform(-20, 20))]
        D = [0, 0, 0, 0, 0] if rng.integers(0, 3) == 0 else [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.01, 0.01)),
                                                              float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-0.5, 0.5))]
        host = []
        for f in range(B):
            keys, desc = gex.download(f)
            host.append((ob.undistort_keypoints(keys, K, D), desc))
        allx = np.concatenate([k["x"] for k, _ in host]); ally = np.concatenate([k["y"] for k, _ in host])
        cut = float(rng.uniform(0, 12))
        bounds = [float(np.floor(allx.min())) + cut, float(np.ceil(allx.max())) - cut, float(np.floor(ally.min())) + cut, float(np.ceil(ally.max())) - cut]
        g = make_grid(bounds[0], bounds[2], bounds[1], bounds[3])
        gp = ob.make_grid_params(bounds[0], bounds[2], bounds[1], bounds[3])
        ratio = float(np.float32(rng.choice([0.7, 0.8, 0.9])))
        m = ORBmatcher(ratio, True, device=0)
        fs = m.frame_set(B, gex.max_keypoints, K, D, g, bounds, sf)
        fs.build_from_extractor(0, gex)
        for f in range(B):
            ku, dd = fs.download(f)
            if ku.tobytes() != host[f][0].tobytes() or dd.tobytes() != host[f][1].tobytes():
                print("DIFFERENCE frame build", dict(round=r, w=w, h=h, nf=nf, B=B, K=K, D=D, frame=f, seed=seed))
                return 1
        counts["frames_built"] += B
        npairs = int(rng.integers(1, B + 1))   # orbm_track_frames takes at most one pair per slot of the set
        cur = rng.integers(0, B, npairs)
        last = rng.integers(0, B, npairs)
        th = float(rng.choice([7.0, 15.0, 30.0])); thd = int(rng.choice([50, 100])); ori = bool(rng.integers(0, 2))
        fs.track(cur, last, th=th, th_dist=thd, nnratio=ratio, check_ori=ori)
        assign, nm = fs.results()
        for p in range(npairs):
            kc, dc = host[cur[p]]
            kl, dl = host[last[p]]
            uvr, lvl, qv = identity_queries(kl, sf, th, bounds)
            start, idx = ob.grid_build(gp, kc)
            wa, _, wn = ob.search_by_projection(4, ratio, ori, thd, uvr, lvl, dl, kl["angle"], qv, None, gp, kc, start, idx, dc,
                                                np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32))
            if nm[p] != wn or not np.array_equal(assign[p, :len(kc)], wa):
                print("DIFFERENCE tracking search", dict(round=r, w=w, h=h, nf=nf, B=B, pair=(int(cur[p]), int(last[p])), th=th, thd=thd, ratio=ratio, ori=ori, seed=seed))
                return 1
        counts["track_pairs"] += npairs
        fs.close()
        del gex
        # ------------------------------------------------------------ stereo
        w, h = shapes[int(rng.integers(0, len(shapes)))]
        nf = int(rng.integers(300, 2000))
        left = synth.make_frames(w, h, 1, stream=int(rng.integers(0, 500)))[0]
        right = np.empty_like(left)
        nb = int(rng.integers(1, 5))
        edges = np.linspace(0, h, nb + 1).astype(int)
        for i in range(nb):
            right[edges[i]:edges[i + 1]] = np.roll(left[edges[i]:edges[i + 1]], -int(rng.integers(0, 60)), axis=1)
        right = np.clip(right.astype(np.int16) + rng.integers(-3, 4, size=right.shape), 0, 255).astype(np.uint8)
        exL = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
        exR = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
        kL, dL = exL(left); kR, dR = exR(right)
        oex = ob.Extractor(nf, 1.2, 8, 20, 7)
        oL, oR = oex(left, want_pyramid=True), oex(right, want_pyramid=True)
        if kL.tobytes() != oL["kps"].tobytes() or kR.tobytes() != oR["kps"].tobytes():
            print("DIFFERENCE stereo extraction", dict(round=r, w=w, h=h, nf=nf, seed=seed))
            return 1
        sfo = oex.scale_factors()
        mbf = float(rng.uniform(20, 400)); mb = mbf / float(rng.uniform(300, 800))
        want_u, want_d, _ = ob.compute_stereo_matches(oL["kps"], oL["desc"], oR["kps"], oR["desc"], levels_of(oex, oL["pyramid"], w, h, 8),
                                                      levels_of(oex, oR["pyramid"], w, h, 8), sfo, (1.0 / sfo).astype(np.float32), mb, mbf)
        got_u, got_d = exL.compute_stereo_matches(exR, mb, mbf)
        if got_u.tobytes() != want_u.tobytes() or got_d.tobytes() != want_d.tobytes():
            print("DIFFERENCE ComputeStereoMatches", dict(round=r, w=w, h=h, nf=nf, mb=mb, mbf=mbf, bands=nb, seed=seed))
            return 1
        counts["stereo_pairs"] += 1
        del exL, exR
        # ------------------------------------------------------------ vocabulary transform
        k, L = int(rng.integers(2, 11)), int(rng.integers(1, 6))
        while k ** L > 200000:
            L -= 1
        voc = make_vocab(rng, k, L, ragged=bool(rng.integers(0, 2)))
        scoring, weighting = int(rng.integers(0, 6)), int(rng.integers(0, 4))
        G = ORBVocabulary(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
        O = ob.Vocabulary(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        for _ in range(3):
            n = int(rng.choice([1, 7, 333, 2000, 4097]))
            desc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            levelsup = int(rng.integers(0, L + 3))
            (gi, gv), (gn, gs, gx) = G.transform(desc, levelsup)
            (oi, ov), (on, os_, ox) = O.transform(desc, levelsup)
            if not (np.array_equal(gi, oi) and gv.tobytes() == ov.tobytes() and np.array_equal(gn, on) and np.array_equal(gs, os_) and np.array_equal(gx, ox)):
                print("DIFFERENCE vocabulary transform", dict(round=r, k=k, L=L, scoring=scoring, weighting=weighting, n=n, levelsup=levelsup, seed=seed))
                return 1
            counts["transforms"] += 1
    print("front-end soak: %d rounds (seed %d): %s -- undistorted keys, grids (through the searches), match tables, mvuRight / mvDepth and "
          "BowVector / FeatureVector all equal to the oracle's; %.0f s" % (rounds, seed, counts, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
