#!/usr/bin/env python3
"""Differential soak of the HIP extractor + stream matcher against the CPU oracle (test infrastructure under tests/):
N random cases of (shape, ORBextractor parameters, image statistics), keypoint records and descriptor bytes compared
byte for byte, then the brute-force match of the case's two frames.  On the GPU box:
    python tests/soak/fuzz_soak.py [cases] [seed] [minside maxw maxh maxarea nfmax [frames-per-call choices, e.g. 8,17,33,64]] > gpurun_out/fuzz_soak.txt
Content kinds: the bench's synthetic scene, white noise, band-limited noise at several scales, checkerboards, ramps +
noise, crops of the photographs in tests/golden/natural.npz, the same clipped / compressed in contrast (saturation and
the minThFAST retry), images with flat halves (empty cells).  Exit code 1 on the first difference (the case is printed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import ORBextractor, ORBmatcher, synth  # noqa: E402


def smooth(a, k):
    if k <= 1:
        return a
    ker = np.ones(k) / k
    a = np.apply_along_axis(lambda r: np.convolve(r, ker, mode="same"), 1, a)
    return np.apply_along_axis(lambda c: np.convolve(c, ker, mode="same"), 0, a)


def make_image(rng, kind, w, h, naturals):
    if kind == "scene":
        return synth.make_frames(w, h, 1, stream=int(rng.integers(0, 1000)))[0]
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "band":
        a = smooth(rng.standard_normal((h, w)), int(rng.choice([2, 3, 5, 9])))
        a = (a - a.min()) / max(float(np.ptp(a)), 1e-9)
        return (a * 255).astype(np.uint8)
    if kind == "checker":
        c = int(rng.integers(2, 24))
        yy, xx = np.mgrid[0:h, 0:w]
        lo, hi = sorted(int(v) for v in rng.integers(0, 256, 2))
        img = np.where(((yy // c) + (xx // c)) % 2 == 0, lo, hi).astype(np.int32)
        return np.clip(img + rng.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)
    if kind == "ramp":
        yy, xx = np.mgrid[0:h, 0:w]
        img = (xx * float(rng.uniform(0.05, 0.4)) + yy * float(rng.uniform(0.05, 0.4))) % 256
        return np.clip(img + rng.normal(0, float(rng.uniform(1, 25)), (h, w)), 0, 255).astype(np.uint8)
    if kind in ("natural", "clipped", "lowcontrast", "halfflat"):
        src = naturals[int(rng.integers(0, len(naturals)))]
        if src.shape[0] < h or src.shape[1] < w:
            reps = (-(-h // src.shape[0]), -(-w // src.shape[1]))
            src = np.tile(src, reps)
        y0, x0 = int(rng.integers(0, src.shape[0] - h + 1)), int(rng.integers(0, src.shape[1] - w + 1))
        img = np.ascontiguousarray(src[y0:y0 + h, x0:x0 + w])
        if kind == "clipped":
            img = np.clip(img.astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)
        elif kind == "lowcontrast":
            img = (img // int(rng.choice([4, 8, 16])) + int(rng.integers(0, 200))).astype(np.uint8)
        elif kind == "halfflat":
            img = img.copy()
            if rng.integers(0, 2):
                img[:, : w // 2] = int(rng.integers(0, 256))
            else:
                img[h // 2:, :] = int(rng.integers(0, 256))
        return img
    raise ValueError(kind)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    # optional: smallest side, largest width / height, largest area (defaults 96, 1400, 720, 700 000; e.g. 34 3000 1700 4000000
    # for tiny frames the reference cannot handle -- refused -- up to 4 Mpx ones that leave the LDS-resident forms)
    minside = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    maxw = int(sys.argv[4]) if len(sys.argv) > 4 else 1400
    maxh = int(sys.argv[5]) if len(sys.argv) > 5 else 720
    maxarea = int(sys.argv[6]) if len(sys.argv) > 6 else 700000
    nfmax = int(sys.argv[7]) if len(sys.argv) > 7 else 4000   # nfeatures below this
    bchoices = [int(x) for x in sys.argv[8].split(',')] if len(sys.argv) > 8 else [2, 2, 2, 3, 4, 6, 9]   # frames per call (sub-batches of >= 8 frames take the XCD-aware grids)
    rng = np.random.default_rng(seed)
    z = np.load(os.path.join(ROOT, "tests", "golden", "natural.npz"))
    naturals = [z[k] for k in z.files if z[k].ndim == 2 and z[k].dtype == np.uint8 and min(z[k].shape) >= 200]
    kinds = ["scene", "noise", "band", "checker", "ramp", "natural", "clipped", "lowcontrast", "halfflat"]
    gm = ORBmatcher(0.7, True, device=0)
    done = refused = 0
    nkp = nmatch = nframes = 0
    per_kind = {k: 0 for k in kinds}
    t0 = time.time()
    while done < ncases:
        w, h = int(rng.integers(minside, maxw)), int(rng.integers(minside, maxh))
        if (w - 32) / max(h - 32, 1) < 0.5 or w * h > maxarea:
            continue
        nf = int(rng.integers(20, nfmax))
        sf = float(np.float32(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, float(rng.uniform(1.05, 1.9))])))
        nl = int(rng.integers(1, 11))
        ini, mn = int(rng.integers(5, 80)), int(rng.integers(1, 30))
        kind = kinds[int(rng.integers(0, len(kinds)))]
        B = int(rng.choice(bchoices))
        try:
            gex = ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h, max_batch=B, device=0)
        except Exception as e:  # shapes the reference would crash on are refused (ORBX_E_UNSUPPORTED)
            if getattr(e, "code", 0) != -5:
                raise
            refused += 1
            continue
        a = make_image(rng, kind, w, h, naturals)
        b = np.roll(a, (int(rng.integers(-3, 4)), int(rng.integers(-6, 7))), (0, 1))   # the "next frame"
        # more frames of OTHER statistics in the same call: calls of more than two frames take the throughput graph
        # (sub-batches on their own streams), one or two the latency chain
        more = [make_image(rng, kinds[int(rng.integers(0, len(kinds)))], w, h, naturals) for _ in range(B - 2)]
        batch = [a, b] + more
        try:
            oex = ob.Extractor(nf, sf, nl, ini, mn)
            refs = [oex(im) for im in batch]
            ra, rb = refs[0], refs[1]
        except RuntimeError:
            refused += 1
            continue
        case = dict(case=done, kind=kind, w=w, h=h, nf=nf, sf=sf, nl=nl, ini=ini, mn=mn, B=B, seed=seed)
        try:
            kps, desc = gex.extract_batch(np.stack(batch))
        except Exception as e:
            print("ERROR in the product path:", e, case)
            return 1
        for tag, (ref, k, d) in enumerate(zip(refs, kps, desc)):
            if len(ref["kps"]) != len(k) or ref["kps"].tobytes() != k.tobytes() or ref["desc"].tobytes() != d.tobytes():
                print("DIFFERENCE frame %d: %d vs %d keypoints" % (tag, len(ref["kps"]), len(k)), case)
                return 1
        if len(ra["kps"]) and len(rb["kps"]):
            m_ref, n_ref = ob.match_bruteforce(rb["desc"], rb["kps"]["angle"], ra["desc"], ra["kps"]["angle"], 0.7, 50, True)
            m_gpu, n_gpu = gm.match_bruteforce(desc[1], kps[1]["angle"], desc[0], kps[0]["angle"])
            if n_ref != n_gpu or not np.array_equal(np.asarray(m_ref), np.asarray(m_gpu)):
                print("DIFFERENCE match: %d vs %d" % (n_ref, n_gpu), case)
                return 1
            nmatch += int(n_gpu)
        nkp += sum(len(k) for k in kps)
        nframes += B
        per_kind[kind] += 1
        done += 1
        del gex
    print("fuzz soak: %d cases (seed %d, %d shapes refused like the reference would crash), %d frames, %d keypoints, %d matches, "
          "all keypoint records, descriptor bytes and match tables equal to the oracle's; %.0f s" % (done, seed, refused, nframes, nkp, nmatch, time.time() - t0))
    print("cases per content kind:", per_kind)
    return 0


if __name__ == "__main__":
    sys.exit(main())
