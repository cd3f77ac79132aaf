#!/usr/bin/env python3
"""Soak of the live modes through examples/multi_robot (the reference's multi-robot main loop on the C ABI): a robot's
stream of results must not depend on HOW it is driven.  Per round a random (mode, robots, shape, features) is run in the
plain form -- a handle per robot, one ticket deep, attached frame set, pinned camera ring -- and then in random other
forms (two tickets deep, detached, pageable frames, one or two latency streams, behind camera hubs of random size and
waiting time); the per-robot checksum (keypoint counts, descriptor words, match-table entries of every frame) and the
means must be equal.  Bit-exactness of these entry points against the oracle is tests/ and tests/soak/fuzz_*.py; this soak is
about concurrency: threads, shared stream pools, hubs.  On the GPU box:
    python tests/soak/fuzz_multi_robot.py [rounds] [seed] > gpurun_out/fuzz_multi_robot.txt"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(exe, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([exe, "--json"] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=e)
    if out.returncode != 0:
        return None, out.stderr[-1500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), ""


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rng = np.random.default_rng(seed)
    import __graft_entry__ as ge
    exe = ge.build_examples()
    shapes = [(640, 480), (1241, 376), (752, 480), (512, 384)]
    nruns = 0
    t0 = time.time()
    for r in range(rounds):
        mode = str(rng.choice(["track", "track", "full", "bf", "extract"]))
        robots = int(rng.integers(1, 9))
        w, h = shapes[int(rng.integers(0, len(shapes)))]
        nfeat = int(rng.choice([500, 1000, 1500, 2000]))
        frames = int(rng.integers(20, 60))
        common = ["--mode", mode, "--robots", robots, "--w", w, "--h", h, "--nfeat", nfeat, "--frames", frames, "--warmup", 4, "--interval", 16]
        base, err = run(exe, common)
        nruns += 1
        if base is None:
            print("ERROR base run", common, err)
            return 1
        for v in range(3):
            extra, env = [], {}
            if mode in ("track", "full") and rng.integers(0, 2):
                extra = ["--hub", int(rng.integers(1, 9)), "--hub-wait", int(rng.choice([0, 40, 500, 2000]))]
            else:
                if mode != "full" and rng.integers(0, 2):
                    extra += ["--depth", 2]
                if rng.integers(0, 2):
                    extra += ["--pinned", 0]
                if mode in ("track", "full") and rng.integers(0, 2):
                    extra += ["--attach", 0]
                if rng.integers(0, 2):
                    env["ORBX_LAT_STREAMS"] = str(rng.choice(["1", "2"]))
            got, err = run(exe, common + extra, env)
            nruns += 1
            if got is None:
                print("ERROR variant run", common, extra, env, err)
                return 1
            if got["checksum"] != base["checksum"] or got["matches_mean"] != base["matches_mean"] or got["keypoints_mean"] != base["keypoints_mean"]:
                print("DIFFERENCE", common, extra, env, {k: (base[k], got[k]) for k in ("checksum", "matches_mean", "keypoints_mean")}, "seed", seed)
                return 1
    print("multi-robot soak: %d rounds (seed %d), %d runs of examples/multi_robot: every robot's results (checksums of keypoint counts, "
          "descriptor words and match tables) independent of how it was driven; %.0f s" % (rounds, seed, nruns, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
