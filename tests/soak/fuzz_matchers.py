#!/usr/bin/env python3
"""Differential soak of the ORBmatcher entry points against the CPU oracle (test infrastructure under tests/ and
tests/soak/fuzz_soak.py): random sizes, thresholds, ratios and seeds through the case generators of tests/matcher_cases.py --
brute force, SearchByBoW (both forms), SearchByProjection modes 3-6, the windowed best (Fuse / SearchBySim3 device part),
SearchForInitialization, SearchForTriangulation, ComputeDistinctiveDescriptors, GetFeaturesInArea.  On the GPU box:
    python tests/soak/fuzz_matchers.py [rounds] [seed] > gpurun_out/fuzz_matchers.txt
Exit code 1 on the first difference (the case is printed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from matcher_cases import make_bow_case, make_init_case, make_proj_case, make_tri_case, noisy_copies  # noqa: E402
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import ORBmatcher, make_grid  # noqa: E402


def rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    big = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # size multiplier of the cases (3: up to ~9 000 features a side)
    rng = np.random.default_rng(seed)
    counts = {}
    t0 = time.time()

    def fail(what, **case):
        print("DIFFERENCE in %s:" % what, case, "seed", seed)
        return 1

    def tally(what, n=1):
        counts[what] = counts.get(what, 0) + n

    for r in range(rounds):
        ratio = float(np.float32(rng.choice([0.6, 0.7, 0.75, 0.8, 0.9, float(rng.uniform(0.5, 0.99))])))
        ori = bool(rng.integers(0, 2))
        # ---- brute force (the stream matcher's rule), sizes across the tile / chunk boundaries
        nq, nt = int(rng.integers(0, 2600 * big)), int(rng.integers(0, 2600 * big))
        if rng.integers(0, 4) == 0:
            nt = int(rng.integers(2040, 5200))
        base = rand_desc(rng, max(nt, 1))[:nt]
        q = noisy_copies(rng, base[rng.integers(0, max(nt, 1), nq)] if nt else rand_desc(rng, nq), int(rng.integers(0, 60))) if nq else np.zeros((0, 32), np.uint8)
        qa, ta = rng.uniform(0, 360, nq).astype(np.float32), rng.uniform(0, 360, nt).astype(np.float32)
        th = int(rng.choice([30, 50, 50, 100, 256]))
        gm = ORBmatcher(ratio, ori, device=0)
        got, n = gm.match_bruteforce(q, qa, base, ta, th)
        want, nw = ob.match_bruteforce(q, qa, base, ta, ratio, th, ori)
        if n != nw or not np.array_equal(got, want):
            return fail("brute force", nq=nq, nt=nt, ratio=ratio, ori=ori, th=th, round=r)
        tally("bruteforce")
        # ---- SearchByBoW, both forms
        nq, nt, nn = int(rng.integers(1, 2400 * big)), int(rng.integers(1, 2400 * big)), int(rng.integers(1, 120 * big))
        c = make_bow_case(rng, nq, nt, nn)
        for by_train in (True, False):
            tv = None if by_train else c["tv"]
            got, n = gm.SearchByBoW(c["qd"], c["qa"], c["qv"], c["qfv"], c["td"], c["ta"], tv, c["tfv"], by_train)
            want, nw = ob.search_by_bow(c["qd"], c["qa"], c["qv"], c["qfv"], c["td"], c["ta"], tv, c["tfv"], ratio, ori, by_train)
            if n != nw or not np.array_equal(got, want):
                return fail("SearchByBoW", nq=nq, nt=nt, nnodes=nn, by_train=by_train, ratio=ratio, ori=ori, round=r)
            tally("bow")
        # ---- SearchByProjection modes 3-6
        nq, nt = int(rng.integers(1, 3200 * big)), int(rng.integers(1, 3200 * big))   # (beyond 8 192 train features the search runs from memory: same results)
        c = make_proj_case(rng, nq, nt)
        g = make_grid(0.0, 0.0, c["w"], c["h"])
        for mode in (3, 4, 5, 6):
            thd = int(rng.choice([50, 64, 100]))
            a0 = np.full(nt, -1, np.int32)
            ga, gocc, gn = gm.SearchByProjection(mode, thd, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], g, c["tk"], c["td"], c["occ"], a0)
            wa, wocc, wn = ob.search_by_projection(mode, ratio, ori, thd, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], c["gp"],
                                                   c["tk"], c["start"], c["idx"], c["td"], c["occ"], a0)
            if gn != wn or not np.array_equal(ga, wa) or not np.array_equal(gocc, wocc):
                return fail("SearchByProjection", mode=mode, nq=nq, nt=nt, th=thd, ratio=ratio, ori=ori, round=r)
            tally("projection")
        # ---- windowed best (Fuse x2 / SearchBySim3 device part)
        inv = (1.0 / (np.float32(1.2) ** np.arange(8)) ** 2).astype(np.float32)
        pred = np.clip(c["lvl"][:, 0] + 1, 0, 7).astype(np.int8)
        for chi2 in (False, True):
            gi, gd = gm.window_best(c["uvr"], pred, c["qd"], c["qv"], g, c["tk"], c["td"], inv, chi2, None, None)
            wi, wd = ob.window_best(c["uvr"], pred, c["qd"], c["qv"], c["gp"], c["tk"], c["start"], c["idx"], c["td"], inv, chi2, None, None)
            if not np.array_equal(gi, wi) or not np.array_equal(gd, wd):
                return fail("window_best", chi2=chi2, nq=nq, nt=nt, round=r)
            tally("window")
        # ---- GetFeaturesInArea
        for _ in range(8):
            x, y, rad = float(rng.uniform(-30, c["w"] + 30)), float(rng.uniform(-30, c["h"] + 30)), float(rng.uniform(1, 150))
            lo, hi = [(-1, -1), (0, 2), (3, 5), (2, -1), (0, -1), (7, 8)][int(rng.integers(0, 6))]
            want = ob.features_in_area(c["gp"], c["tk"], c["start"], c["idx"], x, y, rad, lo, hi)
            got = gm.GetFeaturesInArea(g, c["tk"], x, y, rad, lo, hi)
            if not np.array_equal(got, want):
                return fail("GetFeaturesInArea", x=x, y=y, r=rad, lo=lo, hi=hi, nt=nt, round=r)
            tally("area")
        # ---- SearchForInitialization
        n1, n2 = int(rng.integers(1, 2400 * big)), int(rng.integers(1, 2400 * big))
        ic = make_init_case(rng, n1, n2)
        gi_ = make_grid(0.0, 0.0, ic["w"], ic["h"])
        win = float(rng.choice([30.0, 50.0, 100.0]))
        gmm, gn = gm.SearchForInitialization(ic["q_xy"], win, ic["k1"], ic["d1"], gi_, ic["k2"], ic["d2"])
        wm, wn = ob.search_for_initialization(ic["q_xy"], win, ic["k1"], ic["d1"], ic["gp"], ic["k2"], ic["start"], ic["idx"], ic["d2"], ratio, ori)
        if gn != wn or not np.array_equal(gmm, wm):
            return fail("SearchForInitialization", n1=n1, n2=n2, win=win, ratio=ratio, ori=ori, round=r)
        tally("init")
        # ---- SearchForTriangulation
        n1, n2, nn = int(rng.integers(1, 2200 * big)), int(rng.integers(1, 2200 * big)), int(rng.integers(1, 80 * big))
        tc = make_tri_case(rng, n1, n2, nn)
        cc = tc["c"]
        gmm, gn = gm.SearchForTriangulation(tc["k1"], cc["qd"], 1 - cc["qv"], cc["qfv"], tc["k2"], cc["td"], 1 - cc["tv"], cc["tfv"],
                                            tc["F"], tc["ex"], tc["ey"], tc["sf2"], tc["sigma2"])
        wm, wn = ob.search_for_triangulation(tc["k1"], cc["qd"], 1 - cc["qv"], cc["qfv"], tc["k2"], cc["td"], 1 - cc["tv"], cc["tfv"],
                                             tc["F"], tc["ex"], tc["ey"], tc["sf2"], tc["sigma2"], False, ori)
        if gn != wn or not np.array_equal(gmm, wm):
            return fail("SearchForTriangulation", n1=n1, n2=n2, nnodes=nn, ratio=ratio, ori=ori, round=r)
        tally("triangulation")
        # ---- ComputeDistinctiveDescriptors
        sizes = [0, 1, 2] + rng.integers(1, 90, int(rng.integers(1, 200))).tolist()
        start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        b = rand_desc(rng, len(sizes))
        desc = np.concatenate([noisy_copies(rng, np.repeat(b[i:i + 1], s, axis=0), 20) if s else np.zeros((0, 32), np.uint8) for i, s in enumerate(sizes)])
        if not np.array_equal(gm.ComputeDistinctiveDescriptors(desc, start), ob.distinctive_descriptors(desc, start)):
            return fail("ComputeDistinctiveDescriptors", nobs=len(sizes), round=r)
        tally("distinctive")
    print("matcher soak: %d rounds (seed %d), calls compared with the oracle per entry point: %s -- all match tables, occupancy "
          "tables and counts equal; %.0f s" % (rounds, seed, counts, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
