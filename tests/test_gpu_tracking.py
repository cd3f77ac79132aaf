"""GPU parity of the Tracking-shaped path (round 3): the fused, parallel-resolve SearchByProjection kernel on inputs
built to stress its rounds, the batched Frame construction and the batched frame-to-frame search -- all through the C
ABI, all against the sequential oracle (oracle/orb_match.c follows ORBmatcher.cc:45-129, 1330-1472 query by query)."""
import numpy as np
import pytest

from conftest import random_descriptors
from matcher_cases import make_proj_case, noisy_copies

pytestmark = pytest.mark.gpu

TUM_K = [517.306408, 516.469215, 318.643040, 255.313989]
TUM_D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
KITTI_K = [718.856, 718.856, 607.1928, 185.2157]


def _both(oracle, mode, th, ratio, ori, c, nt):
    from orbslamm_amd import ORBmatcher, make_grid
    g = make_grid(0.0, 0.0, c["w"], c["h"])
    m = ORBmatcher(ratio, ori, device=0)
    a0 = c.get("a0", np.full(nt, -1, np.int32))
    ga, gocc, gn = m.SearchByProjection(mode, th, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], g,
                                        c["tk"], c["td"], c["occ"], a0)
    wa, wocc, wn = oracle.search_by_projection(mode, ratio, ori, th, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"],
                                               c["qo"], c["gp"], c["tk"], c["start"], c["idx"], c["td"], c["occ"], a0)
    stats = m.last_search_stats()
    m.close()
    assert gn == wn, (gn, wn)
    assert np.array_equal(ga, wa) and np.array_equal(gocc, wocc)
    return wn, stats


@pytest.mark.parametrize("mode,th,ratio", [(3, 100, 0.8), (4, 100, 0.9), (5, 64, 0.9), (6, 50, 0.75)])
def test_contended_queries_resolve_in_reference_order(gpu, oracle, mode, th, ratio):
    """Many queries compete for few train features: every query's window holds the same handful of features, so the
    sequential loop's outcome hangs on the order (each accepted query takes a feature away from all later ones).  The
    parallel rounds must reproduce it, including for queries whose MapPoint has no observations (modes 3/4: they take
    nothing away and may be overwritten later)."""
    rng = np.random.default_rng(500 + mode)
    for nq, nt, clusters, spread in ((400, 60, 3, 4.0), (1500, 300, 10, 8.0), (64, 8, 1, 1.0), (3000, 2500, 40, 10.0)):
        c = make_proj_case(rng, nq, nt)
        # train features in tight clusters, queries thrown onto the cluster centres with near-identical descriptors
        cx, cy = rng.uniform(100, 1100, clusters), rng.uniform(50, 330, clusters)
        cl = rng.integers(0, clusters, nt)
        c["tk"]["x"] = (cx[cl] + rng.normal(0, spread, nt)).astype(np.float32)
        c["tk"]["y"] = (cy[cl] + rng.normal(0, spread, nt)).astype(np.float32)
        c["tk"]["octave"] = rng.integers(0, 3, nt)
        base = random_descriptors(rng, clusters)
        c["td"] = noisy_copies(rng, base[cl], 6)
        qc = rng.integers(0, clusters, nq)
        c["qd"] = noisy_copies(rng, base[qc], 6)
        c["uvr"][:, 0] = cx[qc] + rng.normal(0, 1.0, nq)
        c["uvr"][:, 1] = cy[qc] + rng.normal(0, 1.0, nq)
        c["uvr"][:, 2] = 40.0
        c["lvl"][:] = (-1, -1) if mode != 3 else (0, 2)
        c["start"], c["idx"] = oracle.grid_build(c["gp"], c["tk"])
        c["a0"] = rng.integers(-1, 5, nt).astype(np.int32)  # assign is in/out: untouched entries keep their value
        most = 0
        for ori in (True, False):
            for qo in (c["qo"], None, np.zeros(nq, np.uint8)):
                cc = dict(c, qo=qo)
                wn, (rounds, cands) = _both(oracle, mode, th, ratio, ori, cc, nt)
                most = max(most, rounds)
        assert wn > 0 and most > 3  # the case does contend


def test_candidate_list_sizes_lds_arena_and_regrow(gpu, oracle):
    """the candidate list lives in LDS when it fits, else in the handle's arena, which grows once when a call needs
    more than it holds (reported by the kernel, never written past)"""
    rng = np.random.default_rng(77)
    seen = set()
    for nq, nt, r in ((300, 900, 15.0), (600, 2000, 70.0), (2500, 3000, 2000.0), (300, 900, 15.0)):
        c = make_proj_case(rng, nq, nt)
        c["uvr"][:, 2] = r
        if r > 50:
            c["lvl"][:] = (-1, -1)
        # mode 3 lists every feature of the window (the other modes only those within the distance threshold)
        wn, (rounds, cands) = _both(oracle, 3, 100, 0.8, True, c, nt)
        seen.add("lds" if cands <= 16 * nt else ("arena" if cands <= max(64 * nq, 65536) else "regrow"))
        assert wn > 0
    assert seen == {"lds", "arena", "regrow"}, seen


def test_projection_degenerate_shapes(gpu, oracle):
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(5)
    c = make_proj_case(rng, 40, 50)
    # all queries invalid; every train feature occupied on entry; a single query / a single train feature
    _both(oracle, 4, 100, 0.9, True, dict(c, qv=np.zeros(40, np.uint8)), 50)
    _both(oracle, 3, 100, 0.8, True, dict(c, occ=np.ones(50, np.uint8)), 50)
    c1 = make_proj_case(rng, 1, 1)
    _both(oracle, 4, 100, 0.9, True, c1, 1)
    # more queries than the resolve keeps per-query tables for in LDS (a local map's worth of MapPoints): same rounds, from memory
    big_q = make_proj_case(rng, 6000, 1500)
    for mode, th, ratio in ((3, 100, 0.8), (4, 100, 0.9)):
        wn, _ = _both(oracle, mode, th, ratio, True, big_q, 1500)
        assert wn > 500
    m = ORBmatcher(0.9, True, device=0)
    g = make_grid(0.0, 0.0, 1241.0, 376.0)
    a, occ, n = m.SearchByProjection(4, 100, np.zeros((0, 3), np.float32), np.zeros((0, 2), np.int8), np.zeros((0, 32), np.uint8),
                                     np.zeros(0, np.float32), None, None, g, c["tk"], c["td"], c["occ"], np.full(50, -1, np.int32))
    assert n == 0 and (a == -1).all() and np.array_equal(occ, c["occ"])
    # (10 001 train features -- beyond the LDS form -- used to be refused here; since round 6 they are searched from memory:
    # tests/test_gpu_large_frames.py)
    big = make_proj_case(rng, 10, 10001)
    _both(oracle, 4, 100, 0.9, True, big, 10001)


def _identity_queries(keys_un, sf, th, bounds):
    """the queries orbm_track_frames derives on the device (ORBmatcher.cc:1375-1392 with the identity pose)"""
    uvr = np.stack([keys_un["x"], keys_un["y"], (np.float32(th) * sf[keys_un["octave"]]).astype(np.float32)], axis=1).astype(np.float32)
    lvl = np.stack([keys_un["octave"] - 1, keys_un["octave"] + 1], axis=1).astype(np.int8)
    x, y = keys_un["x"], keys_un["y"]
    qv = ~((x < bounds[0]) | (x > bounds[1]) | (y < bounds[2]) | (y > bounds[3]))
    return uvr, lvl, qv.astype(np.uint8)


@pytest.mark.parametrize("w,h,nf,K,D", [(640, 480, 1000, TUM_K, TUM_D), (1241, 376, 2000, KITTI_K, [0, 0, 0, 0, 0])])
def test_frame_set_and_batched_tracking_search(gpu, oracle, w, h, nf, K, D):
    """extractor -> frame set -> SearchByProjection(Cur, Last) for a batch of consecutive frames without leaving HBM:
    mvKeysUn, descriptors and every match table equal the oracle's, fed with the downloaded arrays"""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    B = 6
    fr = synth.make_frames(w, h, B, stream=5)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    gex.extract_batch_device(*gex.upload_frames(fr))
    host = []
    for f in range(B):
        keys, desc = gex.download(f)
        host.append((oracle.undistort_keypoints(keys, K, D), desc))
    allx = np.concatenate([k["x"] for k, _ in host]); ally = np.concatenate([k["y"] for k, _ in host])
    # any box works as "image bounds" (Frame::ComputeImageBounds is the caller's): one that cuts a few keypoints off
    bounds = [float(np.floor(allx.min())) + 6, float(np.ceil(allx.max())) - 6, float(np.floor(ally.min())) + 6, float(np.ceil(ally.max())) - 6]
    g = make_grid(bounds[0], bounds[2], bounds[1], bounds[3])
    gp = oracle.make_grid_params(bounds[0], bounds[2], bounds[1], bounds[3])
    m = ORBmatcher(0.9, True, device=0)
    cap = gex.max_keypoints
    fs = m.frame_set(B, cap, K, D, g, bounds, sf)
    fs.build_from_extractor(0, gex)
    for f in range(B):
        ku, dd = fs.download(f)
        assert ku.tobytes() == host[f][0].tobytes() and dd.tobytes() == host[f][1].tobytes()
    cur = np.arange(1, B)
    last = np.arange(0, B - 1)
    for th, ori in ((15.0, True), (30.0, True), (15.0, False)):
        fs.track(cur, last, th=th, th_dist=100, nnratio=0.9, check_ori=ori)
        assign, nm = fs.results()
        total = 0
        for p in range(B - 1):
            kc, dc = host[cur[p]]
            kl, dl = host[last[p]]
            uvr, lvl, qv = _identity_queries(kl, sf, th, bounds)
            start, idx = oracle.grid_build(gp, kc)
            wa, _, wn = oracle.search_by_projection(4, 0.9, ori, 100, uvr, lvl, dl, kl["angle"], qv, None, gp, kc, start, idx, dc,
                                                    np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32))
            assert nm[p] == wn, (p, nm[p], wn)
            assert np.array_equal(assign[p, :len(kc)], wa)
            total += wn
            r, cands = fs.stats(p)
            assert r >= 1 and cands > 0
        assert total > 200 * (B - 1)
    # a frame set built slot by slot from device pointers gives the same frames (ring order: slot0 wraps around)
    fs2 = m.frame_set(4, cap, K, D, g, bounds, sf)
    dk, dd, dc, rcap = gex.device_results()
    fs2.build(3, 2, dk + 2 * rcap * 28, dd + 2 * rcap * 32, dc + 2 * 4, rcap)   # frames 2, 3 -> slots 3, 0
    for slot, f in ((3, 2), (0, 3)):
        ku, de = fs2.download(slot)
        assert ku.tobytes() == host[f][0].tobytes() and de.tobytes() == host[f][1].tobytes()
    fs2.track([0], [3], th=15.0)
    a2, n2 = fs2.results()
    fs.track([3], [2], th=15.0)
    a1, n1 = fs.results()
    assert n1[0] == n2[0] and np.array_equal(a1[0, :len(host[3][0])], a2[0, :len(host[3][0])])
    fs.close(); fs2.close()


def test_frame_set_bow_and_batched_search_by_bow(gpu, oracle):
    """Frame::ComputeBoW for a batch of slots and SearchByBoW(KeyFrame, Frame) for a batch of slot pairs, all in HBM:
    BowVectors (ids and float64 values, bit for bit), and every match table, against the oracle fed with the downloads"""
    from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth
    from vocab_cases import make_vocab
    w, h, nf, B = 640, 480, 1000, 5
    rng = np.random.default_rng(17)
    for (k, L, levelsup), voc in (((10, 4, 2), make_vocab(rng, 10, 4)), ((9, 3, 4), make_vocab(rng, 9, 3, ragged=False))):
        G = ORBVocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
        O = oracle.Vocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        fr = synth.make_frames(w, h, B, stream=9)
        gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
        gex.extract_batch_device(*gex.upload_frames(fr))
        host = [gex.download(f) for f in range(B)]
        m = ORBmatcher(0.7, True, device=0)
        g = make_grid(0.0, 0.0, float(w), float(h))
        fs = m.frame_set(B + 2, gex.max_keypoints, TUM_K, [0, 0, 0, 0, 0], g, [0.0, float(w), 0.0, float(h)], np.array(gex.GetScaleFactors(), np.float32))
        fs.build_from_extractor(1, gex)            # frames 0..B-1 -> slots 1..B
        fs.compute_bow(G, 1, B, levelsup)
        fvs = []
        for f in range(B):
            keys, desc = host[f]
            (owid, owval), ofv = O.transform(desc, levelsup)
            wid, wval = fs.bow_vector(1 + f)
            assert np.array_equal(wid, owid) and wval.tobytes() == owval.tobytes()
            fvs.append(ofv)
        kf = np.array([1, 2, 3, 4, 1])        # KeyFrame slots (the last pair: two frames apart)
        cur = np.array([2, 3, 4, 5, 3])
        for ratio, ori in ((0.7, True), (0.9, False)):
            fs.search_by_bow(kf, cur, nnratio=ratio, check_ori=ori)
            match, nm = fs.bow_results()
            total = 0
            for p in range(len(kf)):
                (kq, dq), (kt, dt) = host[kf[p] - 1], host[cur[p] - 1]
                want, wn = oracle.search_by_bow(dq, kq["angle"], None, fvs[kf[p] - 1], dt, kt["angle"], None, fvs[cur[p] - 1], ratio, ori, True)
                assert nm[p] == wn, (p, nm[p], wn)
                assert np.array_equal(match[p, :len(kt)], want)
                total += wn
            assert total > 100
        # an empty slot (never built) has no words and matches nothing
        fs.compute_bow(G, 0, 1, levelsup)
        assert len(fs.bow_vector(0)[0]) == 0
        fs.search_by_bow([0], [1], 0.7, True)
        assert fs.bow_results()[1][0] == 0
        fs.close(); m.close(); G.close()


@pytest.mark.parametrize("lat_streams", ["", "1", "2"])
@pytest.mark.parametrize("w,h,nf,K,D,P", [(1241, 376, 2000, KITTI_K, [0, 0, 0, 0, 0], 1), (640, 480, 1000, TUM_K, TUM_D, 2)])
def test_live_stream_chain_attached_to_the_extractor(gpu, oracle, monkeypatch, lat_streams, w, h, nf, K, D, P):
    """The live stream (round 4): one host frame per call (P = 2: two cameras' frames per call), the frame set attached
    to the extractor -- upload, extraction, Frame tail and SearchByProjection(Cur, Last) are ONE chain on the
    extractor's stream, keypoints / descriptors come back behind the ticket's flag and the match table behind the
    resolve kernel's flag.  Every frame's keypoints, descriptors and match table against the oracle, over enough
    frames that result sets, slots and tickets wrap around; both forms of the latency chain (one queue / two)."""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    if lat_streams:
        monkeypatch.setenv("ORBX_LAT_STREAMS", lat_streams)
    else:
        monkeypatch.delenv("ORBX_LAT_STREAMS", raising=False)
    nfr = 7
    cams = [synth.make_frames(w, h, nfr, stream=20 + j) for j in range(P)]
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=P, device=0)
    oex = oracle.Extractor(nf, 1.2, 8, 20, 7)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    bounds = [0.0, float(w), 0.0, float(h)]
    g = make_grid(*[bounds[i] for i in (0, 2, 1, 3)])
    gp = oracle.make_grid_params(bounds[0], bounds[2], bounds[1], bounds[3])
    m = ORBmatcher(0.9, True, device=0)
    fs = m.frame_set(4 * P, gex.max_keypoints, K, D, g, bounds, sf)
    fs.attach(gex)
    prev = [None] * P
    matched = 0
    for i in range(nfr):
        batch = np.stack([cams[j][i] for j in range(P)])
        tk = gex.submit_host(batch, match=False)
        cur = [(i & 3) * P + j for j in range(P)]
        last = [((i - 1) & 3) * P + j for j in range(P)]
        fs.build_from_extractor(cur[0], gex)
        if i:
            fs.track(cur, last, th=15.0)
        kps, desc, n, _, _ = gex.collect_host(tk, view=False)
        if i:
            assign, nm = fs.results()
        for j in range(P):
            ref = oex(cams[j][i])
            assert n[j] == len(ref["kps"]) and kps[j, :n[j]].tobytes() == ref["kps"].tobytes() and desc[j, :n[j]].tobytes() == ref["desc"].tobytes()
            ku = oracle.undistort_keypoints(ref["kps"], K, D)
            if i:
                kl, dl = prev[j]
                uvr, lvl, qv = _identity_queries(kl, sf, 15.0, bounds)
                start, idx = oracle.grid_build(gp, ku)
                wa, _, wn = oracle.search_by_projection(4, 0.9, True, 100, uvr, lvl, dl, kl["angle"], qv, None, gp, ku, start, idx, ref["desc"],
                                                        np.zeros(len(ku), np.uint8), np.full(len(ku), -1, np.int32))
                assert nm[j] == wn and np.array_equal(assign[j, :len(ku)], wa), (i, j, nm[j], wn)
                matched += wn
            prev[j] = (ku, ref["desc"])
    assert matched > 200 * P * (nfr - 1)
    # detached again the set still holds its frames and searches on the matcher's own stream
    fs.attach(None)
    fs.track([((nfr - 1) & 3) * P], [((nfr - 2) & 3) * P], th=15.0)
    a2, n2 = fs.results()
    nlast = len(prev[0][0])
    assert n2[0] == nm[0] and np.array_equal(a2[0, :nlast], assign[0, :nlast])
    fs.close(); m.close(); gex.close()


def test_projection_search_at_the_largest_frame_capacity(gpu, oracle):
    """at cap = 8192 (kProjMaxTrain) the per-train tables alone take 128 KB of the resolve's LDS -- the plan
    must shrink the per-query tables / candidate lists (rounds from memory) instead of asking for more LDS than a CU has"""
    rng = np.random.default_rng(8192)
    for nq, nt in ((4096, 8192), (6000, 7000)):
        c = make_proj_case(rng, nq, nt)
        for mode, th, ratio in ((4, 100, 0.9), (3, 100, 0.8)):
            wn, _ = _both(oracle, mode, th, ratio, True, c, nt)
            assert wn > 300
    from orbslamm_amd import ORBmatcher, make_grid
    m = ORBmatcher(0.9, True, device=0)
    g = make_grid(0.0, 0.0, 1241.0, 376.0)
    fs = m.frame_set(2, 8192, KITTI_K, [0, 0, 0, 0, 0], g, [0.0, 1241.0, 0.0, 376.0], np.ones(8, np.float32))
    fs.track([0], [1], th=15.0)       # two empty slots of the largest capacity: the launch itself must be accepted
    a, n = fs.results()
    assert n[0] == 0
    fs.close(); m.close()


def local_map_queries(rng, frames, sf, th, nq_target):
    """a local map's worth of projected MapPoints made from the keypoints of neighbouring frames: position jittered by a
    pixel or two, mnTrackScaleLevel = the keypoint's octave, viewCos on either side of 0.998 (RadiusByViewingCos,
    ORBmatcher.cc:131-137), its descriptor, a few MapPoints without observations"""
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


@pytest.mark.parametrize("attached", [False, True])
def test_search_local_points_against_a_resident_frame(gpu, oracle, attached):
    """Tracking::SearchLocalPoints' search (mode 3, ORBmatcher.cc:45-129) through the frame-set entry: the frame stays in
    HBM, 3 000 local MapPoints go up in one pinned block; th = 1 and th = 3, with and without features already taken"""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    w, h, nf = 1241, 376, 2000
    fr = synth.make_frames(w, h, 3, stream=31)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1 if attached else 3, device=0)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    bounds = [0.0, float(w), 0.0, float(h)]
    g = make_grid(0.0, 0.0, float(w), float(h))
    gp = oracle.make_grid_params(0.0, 0.0, float(w), float(h))
    m = ORBmatcher(0.8, True, device=0)
    fs = m.frame_set(4, gex.max_keypoints, KITTI_K, [0, 0, 0, 0, 0], g, bounds, sf)
    host = []
    if attached:
        fs.attach(gex)
        for f in range(3):
            tk = gex.submit_host(fr[f][None], match=False)
            fs.build_from_extractor(f, gex)
            kps, desc, n, _, _ = gex.collect_host(tk, view=False)
            host.append((kps[0, :n[0]].copy(), desc[0, :n[0]].copy()))
    else:
        gex.extract_batch_device(*gex.upload_frames(fr))
        fs.build_from_extractor(0, gex)
        host = [gex.download(f) for f in range(3)]
    rng = np.random.default_rng(303)
    kc, dc = host[2]
    start, idx = oracle.grid_build(gp, kc)
    total = 0
    for th, nq, with_occ in ((1.0, 3000, False), (3.0, 3000, True), (1.0, 4000, True), (3.0, 500, False)):
        uvr, ql, qd, qv, qo = local_map_queries(rng, host[:2], sf, th, nq)
        occ = (rng.random(len(kc)) < 0.3).astype(np.uint8) if with_occ else np.zeros(len(kc), np.uint8)
        fs.track_local_points(2, uvr, ql, qd, qv, qo, occ if with_occ else None)
        assign, nm = fs.results()
        wa, _, wn = oracle.search_by_projection(3, 0.8, True, 100, uvr, ql, qd, None, qv, qo, gp, kc, start, idx, dc, occ, np.full(len(kc), -1, np.int32))
        assert nm[0] == wn and np.array_equal(assign[0, :len(kc)], wa), (th, nq, nm[0], wn)
        total += wn
    assert total > 1500
    fs.close(); m.close(); gex.close()


def test_track_with_projections_from_the_host(gpu, oracle):
    """TrackWithMotionModel's search with a REAL pose (orbm_track_frame_projected): LastFrame's features stay in HBM, the
    caller's projections of them (here: a small similarity -- shift, in-plane rotation, scale -- applied to the undistorted
    keypoints, radius th * scale[octave], some MapPoints missing / behind the camera / without observations) go up; mode 4
    with and without the rotation check, mode 5 (the relocalisation search) and features already taken"""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    w, h, nf = 1241, 376, 2000
    fr = synth.make_frames(w, h, 2, stream=41)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, device=0)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    bounds = [0.0, float(w), 0.0, float(h)]
    g = make_grid(0.0, 0.0, float(w), float(h))
    gp = oracle.make_grid_params(0.0, 0.0, float(w), float(h))
    m = ORBmatcher(0.9, True, device=0)
    fs = m.frame_set(2, gex.max_keypoints, KITTI_K, [0, 0, 0, 0, 0], g, bounds, sf)
    gex.extract_batch_device(*gex.upload_frames(fr))
    fs.build_from_extractor(0, gex)
    (kl, dl), (kc, dc) = gex.download(0), gex.download(1)
    start, idx = oracle.grid_build(gp, kc)
    rng = np.random.default_rng(4141)
    total = 0
    for mode, th, ori, thd, with_occ in ((4, 15.0, True, 100, False), (4, 7.0, False, 100, True), (5, 10.0, True, 64, False)):
        a = np.deg2rad(0.4)
        u = (np.cos(a) * (kl["x"] - 600) - np.sin(a) * (kl["y"] - 180)) * 1.004 + 600 + 1.5
        v = (np.sin(a) * (kl["x"] - 600) + np.cos(a) * (kl["y"] - 180)) * 1.004 + 180 - 0.8
        uvr = np.stack([u, v, np.float32(th) * sf[kl["octave"]]], axis=1).astype(np.float32)
        lvl = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
        qv = ((rng.random(len(kl)) < 0.85) & (u >= 0) & (u <= w) & (v >= 0) & (v <= h)).astype(np.uint8)
        qo = (rng.random(len(kl)) < 0.9).astype(np.uint8)
        occ = (rng.random(len(kc)) < 0.2).astype(np.uint8) if with_occ else np.zeros(len(kc), np.uint8)
        fs.track_projected(1, 0, uvr, lvl, qv, qo, occ if with_occ else None, th_dist=thd, nnratio=0.9, check_ori=ori, mode=mode)
        assign, nm = fs.results()
        wa, _, wn = oracle.search_by_projection(mode, 0.9, ori, thd, uvr, lvl, dl, kl["angle"], qv, qo, gp, kc, start, idx, dc, occ, np.full(len(kc), -1, np.int32))
        assert nm[0] == wn and np.array_equal(assign[0, :len(kc)], wa), (mode, th, nm[0], wn)
        total += wn
    assert total > 1500
    fs.close(); m.close(); gex.close()


def test_frame_to_frame_search_regrows_its_candidate_arena(gpu, oracle):
    """A small, dense frame (2 400 features on 401 x 263) searched against itself with a wide window (th = 60) lists far
    more than the arena's 64 candidates per feature: orbm_track_results grows the arena and runs the search again (it used to
    return ORBX_E_CAPACITY -- found by tests/soak/fuzz_frontend.py); the table equals the sequential oracle's."""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    w, h, nf, B = 401, 263, 2400, 2
    # a periodic texture: every corner looks like every other, so all the spatial neighbours of a query survive the Hamming
    # threshold and are listed
    yy, xx = np.mgrid[0:h + 2, 0:w + 3]
    base = np.where(((yy // 7) + (xx // 7)) % 2 == 0, 40, 210) + np.random.default_rng(12).integers(-2, 3, yy.shape)
    base = np.clip(base, 0, 255).astype(np.uint8)
    fr = np.stack([np.ascontiguousarray(base[:h, :w]), np.ascontiguousarray(base[2:, 3:])])
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    gex.extract_batch_device(*gex.upload_frames(fr))
    K, D = [400.0, 400.0, 200.0, 131.0], [0, 0, 0, 0, 0]
    host = []
    for f in range(B):
        keys, desc = gex.download(f)
        host.append((oracle.undistort_keypoints(keys, K, D), desc))
    bounds = [0.0, float(w), 0.0, float(h)]
    g = make_grid(bounds[0], bounds[2], bounds[1], bounds[3])
    gp = oracle.make_grid_params(bounds[0], bounds[2], bounds[1], bounds[3])
    m = ORBmatcher(0.9, True, device=0)
    fs = m.frame_set(B, gex.max_keypoints, K, D, g, bounds, sf)
    fs.build_from_extractor(0, gex)
    for rep in range(2):   # the second call runs in the grown arena without the retry
        fs.track([0, 1], [0, 0], th=60.0, th_dist=100, nnratio=0.9, check_ori=True)
        assign, nm = fs.results()
        for p, (c, l) in enumerate(((0, 0), (1, 0))):
            kc, dc = host[c]
            kl, dl = host[l]
            uvr, lvl, qv = _identity_queries(kl, sf, 60.0, bounds)
            start, idx = oracle.grid_build(gp, kc)
            wa, _, wn = oracle.search_by_projection(4, 0.9, True, 100, uvr, lvl, dl, kl["angle"], qv, None, gp, kc, start, idx, dc,
                                                    np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32))
            assert nm[p] == wn and np.array_equal(assign[p, :len(kc)], wa)
        r, cands = fs.stats(0)
        assert cands > 64 * gex.max_keypoints   # the scenario does overflow the initial arena
    fs.close()


def test_search_by_bow_on_the_set_searches_the_root_node_of_a_ragged_tree(gpu, oracle):
    """A word that is a leaf ABOVE the FeatureVector's level keeps node id 0 (DBoW2: `*nid = 0`), so a frame can hold one node
    more than the level has vocabulary nodes.  The batched search launched one workgroup per node of the LEVEL and never
    searched the node with the highest id (found by tests/soak/fuzz_tracking.py).  Hand-built tree: k = 3, L = 3, one level-1 node
    is a leaf; at levelsup = 1 the FeatureVectors have the six level-2 nodes + node 0."""
    from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth
    rng = np.random.default_rng(5)
    parent = [0, 0, 0] + [2] * 3 + [3] * 3 + sum([[p] * 3 for p in range(4, 10)], [])
    is_leaf = [1, 0, 0] + [0] * 6 + [1] * 18
    desc = [None] * len(parent)
    root = rng.integers(0, 256, 32, dtype=np.uint8)
    for i, p in enumerate(parent):
        d = (root if p == 0 else desc[p - 1]).copy()
        for b in rng.integers(0, 256, 40):
            d[b // 8] ^= np.uint8(1 << (b % 8))
        desc[i] = d
    voc = dict(parent=np.array(parent, np.int32), is_leaf=np.array(is_leaf, np.uint8), desc=np.stack(desc),
               weight=rng.uniform(0.5, 3.0, len(parent)).astype(np.float64))
    k, L, levelsup = 3, 3, 1
    G = ORBVocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
    O = oracle.Vocabulary(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    w, h, nf, B = 640, 480, 1000, 3
    fr = synth.make_frames(w, h, B, stream=14)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, device=0)
    gex.extract_batch_device(*gex.upload_frames(fr))
    host = [gex.download(f) for f in range(B)]
    m = ORBmatcher(0.9, False, device=0)
    g = make_grid(0.0, 0.0, float(w), float(h))
    fs = m.frame_set(B, gex.max_keypoints, TUM_K, [0, 0, 0, 0, 0], g, [0.0, float(w), 0.0, float(h)], np.array(gex.GetScaleFactors(), np.float32))
    fs.build_from_extractor(0, gex)
    fs.compute_bow(G, 0, B, levelsup)
    fvs = [O.transform(host[f][1], levelsup)[1] for f in range(B)]
    assert all(len(fv[0]) == 7 and fv[0][0] == 0 for fv in fvs)   # six level-2 nodes + the root's
    kf, cur = [0, 1, 2], [1, 2, 0]
    fs.search_by_bow(kf, cur, nnratio=0.9, check_ori=False)
    match, nm = fs.bow_results()
    in_last_node = 0
    for p in range(3):
        (kq, dq), (kt, dt) = host[kf[p]], host[cur[p]]
        want, wn = oracle.search_by_bow(dq, kq["angle"], None, fvs[kf[p]], dt, kt["angle"], None, fvs[cur[p]], 0.9, False, True)
        assert nm[p] == wn and np.array_equal(match[p, :len(kt)], want)
        node, start, idx = fvs[kf[p]]
        last = set(idx[start[-2]:start[-1]].tolist())   # the KeyFrame features of the node with the highest id
        in_last_node += sum(1 for q in want if q in last)
    assert in_last_node > 0   # the scenario has matches that only a search of that node finds
    fs.close(); m.close(); G.close(); gex.close()
