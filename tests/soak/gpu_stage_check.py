#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP extractor/matcher against the CPU oracle on
the GPU box.  Diagnostic tool (not a test): prints where the first divergence is."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as ob  # noqa: E402
from orbslamm_amd import ORBextractor, ORBmatcher, synth, unpack_candidates  # noqa: E402


def check_frame(w, h, nfeat, B=2, verbose=True):
    print("=== %dx%d nfeatures=%d B=%d" % (w, h, nfeat, B))
    frames = synth.make_frames(w, h, B)
    oex = ob.Extractor(nfeat, 1.2, 8, 20, 7)
    gex = ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    t = time.time()
    kps, desc = gex.extract_batch(frames)
    print("gpu extract_batch %.1f ms" % ((time.time() - t) * 1e3))
    ok_all = True
    for f in range(B):
        ref = oex(frames[f], want_pyramid=True)
        # pyramid
        off = 0
        for l in range(8):
            lw, lh = oex.level_size(w, h, l)
            rl = ref["pyramid"][off:off + lw * lh].reshape(lh, lw)
            off += lw * lh
            gl = gex.pyramid_level(f, l)
            if gl.shape != rl.shape or not np.array_equal(gl, rl):
                nd = int((gl != rl).sum()) if gl.shape == rl.shape else -1
                print("  frame %d level %d PYRAMID MISMATCH (%d px differ) shapes %s %s" % (f, l, nd, gl.shape, rl.shape))
                ok_all = False
            gb = gex.pyramid_level(f, l, blurred=True)
            rb = ob.gaussian7(rl)
            if not np.array_equal(gb, rb):
                print("  frame %d level %d BLUR MISMATCH (%d px differ)" % (f, l, int((gb != rb).sum())))
                ok_all = False
            # FAST candidates
            rc = oex.level_candidates(rl)
            gc = gex.level_candidates(f, l)
            gx, gy, gr, go = unpack_candidates(gc)
            order = np.argsort(go, kind="stable")
            gx, gy, gr = gx[order], gy[order], gr[order]
            same = len(rc) == len(gx) and np.array_equal(rc["x"], gx) and np.array_equal(rc["y"], gy) and np.array_equal(rc["score"], gr)
            if not same:
                print("  frame %d level %d FAST MISMATCH: oracle %d gpu %d" % (f, l, len(rc), len(gx)))
                sr = set(zip(rc["x"].tolist(), rc["y"].tolist(), rc["score"].tolist()))
                sg = set(zip(gx.tolist(), gy.tolist(), gr.tolist()))
                print("     only oracle:", sorted(sr - sg)[:8], " only gpu:", sorted(sg - sr)[:8])
                ok_all = False
        rk, rd = ref["kps"], ref["desc"]
        gk, gd = kps[f], desc[f]
        print("  frame %d: oracle %d kps (%s), gpu %d kps" % (f, len(rk), ref["kept_counts"].tolist(), len(gk)))
        if len(rk) != len(gk):
            ok_all = False
            continue
        for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            if not np.array_equal(rk[name], gk[name]):
                bad = np.nonzero(rk[name] != gk[name])[0]
                print("    field %s differs at %d keypoints, first %s: oracle %s gpu %s" % (
                    name, len(bad), bad[:5], rk[name][bad[:5]], gk[name][bad[:5]]))
                ok_all = False
        if not np.array_equal(rd, gd):
            bad = np.nonzero((rd != gd).any(axis=1))[0]
            print("    descriptors differ at %d keypoints, first %s" % (len(bad), bad[:8]))
            ok_all = False
    print("  => %s" % ("BIT-EXACT" if ok_all else "MISMATCH"))
    return ok_all, frames, kps, desc


def check_match(kps, desc):
    gm = ORBmatcher(0.7, True)
    m_ref, n_ref = ob.match_bruteforce(desc[1], kps[1]["angle"], desc[0], kps[0]["angle"], 0.7, 50, True)
    m_gpu, n_gpu = gm.match_bruteforce(desc[1], kps[1]["angle"], desc[0], kps[0]["angle"])
    ok = n_ref == n_gpu and np.array_equal(m_ref, m_gpu)
    print("match bruteforce: oracle %d gpu %d  => %s" % (n_ref, n_gpu, "BIT-EXACT" if ok else "MISMATCH"))
    if not ok:
        bad = np.nonzero(m_ref != m_gpu)[0]
        print("   differ at", len(bad), bad[:10], m_ref[bad[:10]], m_gpu[bad[:10]])
    return ok


if __name__ == "__main__":
    allok = True
    ok, fr, kps, desc = check_frame(640, 480, 1000)
    allok &= ok
    allok &= check_match(kps, desc)
    ok, fr, kps, desc = check_frame(1241, 376, 2000)
    allok &= ok
    allok &= check_match(kps, desc)
    print("ALL OK" if allok else "SOME MISMATCH")
    sys.exit(0 if allok else 1)
