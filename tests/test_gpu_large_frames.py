"""Frames of more than 8 192 features: the reference's loops take any number of features per frame
(Frame.cc:327-380, ORBmatcher.cc:45-129, TemplatedVocabulary.h:1120-1160; Tracking.cc:126 already doubles nFeatures for
the initialisation), the LDS forms of the projection search, of Frame::Frame's tail and of ORBVocabulary::transform hold
8 192.  Beyond that the library takes its memory-resident forms (ProjCommon::big, frame_build_big,
k_voc_aggregate_mem) -- same lists, same rounds, slower -- up to 65 535 features per frame, the width of a feature index
in a candidate entry.  Everything here runs ABOVE the old ceiling and is held to the oracle bit for bit."""
import numpy as np
import pytest

from matcher_cases import make_proj_case, noisy_copies
from test_gpu_tracking import _both, _identity_queries, local_map_queries

pytestmark = pytest.mark.gpu

K_BIG = [718.856, 718.856, 960.0, 540.0]


@pytest.mark.parametrize("mode,th,ratio", [(3, 100, 0.8), (4, 100, 0.9), (5, 64, 0.9), (6, 50, 0.75)])
def test_projection_search_beyond_the_lds_form(gpu, oracle, mode, th, ratio):
    """the host-array entry (what the drop-in ORBmatcher members call): 12 000 and 30 000 train features, every mode"""
    rng = np.random.default_rng(9000 + mode)
    for nq, nt, w, h in ((5000, 12000, 1920.0, 1080.0), (20000, 30000, 3000.0, 1700.0)):
        c = make_proj_case(rng, nq, nt, w, h)
        wn, _ = _both(oracle, mode, th, ratio, True, c, nt)
        assert wn > nq // 20


def test_projection_search_refuses_what_its_entries_cannot_index(gpu, oracle):
    """a candidate entry holds the train index in 16 bits: 65 535 features pass, 65 537 are refused -- never truncated"""
    from orbslamm_amd import ORBmatcher, make_grid
    from orbslamm_amd._lib import OrbError
    rng = np.random.default_rng(65535)
    c = make_proj_case(rng, 3000, 65535, 4000.0, 3000.0)
    wn, _ = _both(oracle, 4, 100, 0.9, True, c, 65535)
    assert wn > 100
    big = make_proj_case(rng, 10, 65537, 4000.0, 3000.0)
    m = ORBmatcher(0.9, True, device=0)
    with pytest.raises(OrbError):
        m.SearchByProjection(4, 100, big["uvr"], big["lvl"], big["qd"], big["qa"], None, None, make_grid(0.0, 0.0, 4000.0, 3000.0),
                             big["tk"], big["td"], big["occ"], np.full(65537, -1, np.int32))
    m.close()


def _big_extraction(oracle, nframes=3, nf=12000, w=1920, h=1080, stream=77):
    from orbslamm_amd import ORBextractor, synth
    fr = synth.make_frames(w, h, nframes, stream=stream)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nframes, device=0)
    gex.extract_batch_device(*gex.upload_frames(fr))
    host = [gex.download(f) for f in range(nframes)]
    assert min(len(k) for k, _ in host) > 8192, [len(k) for k, _ in host]   # above the LDS forms' ceiling, or the test tests nothing
    return gex, host


def test_frame_set_of_large_frames(gpu, oracle):
    """extractor (12 000 features on 1920 x 1080) -> frame set -> SearchByProjection(Cur, Last) for consecutive frames,
    SearchLocalPoints' search and TrackWithMotionModel's search with the caller's projections: frames built by the
    general-size path, grids walked from memory, the resolve's tables in memory"""
    from orbslamm_amd import ORBmatcher, make_grid
    w, h = 1920, 1080
    gex, host = _big_extraction(oracle)
    sf = np.array(gex.GetScaleFactors(), np.float32)
    bounds = [6.0, float(w) - 6, 6.0, float(h) - 6]
    g = make_grid(bounds[0], bounds[2], bounds[1], bounds[3])
    gp = oracle.make_grid_params(bounds[0], bounds[2], bounds[1], bounds[3])
    m = ORBmatcher(0.9, True, device=0)
    fs = m.frame_set(3, gex.max_keypoints, K_BIG, [0, 0, 0, 0, 0], g, bounds, sf)
    fs.build_from_extractor(0, gex)
    for f in range(3):
        ku, dd = fs.download(f)
        assert ku.tobytes() == host[f][0].tobytes() and dd.tobytes() == host[f][1].tobytes()
    # frame to frame, identity pose (orbm_track_frames)
    for th, ori in ((15.0, True), (7.0, False)):
        fs.track([1, 2], [0, 1], th=th, th_dist=100, nnratio=0.9, check_ori=ori)
        assign, nm = fs.results()
        for p, (cur, last) in enumerate(((1, 0), (2, 1))):
            kc, dc = host[cur]
            kl, dl = host[last]
            uvr, lvl, qv = _identity_queries(kl, sf, th, bounds)
            start, idx = oracle.grid_build(gp, kc)
            wa, _, wn = oracle.search_by_projection(4, 0.9, ori, 100, uvr, lvl, dl, kl["angle"], qv, None, gp, kc, start, idx, dc,
                                                    np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32))
            assert nm[p] == wn and np.array_equal(assign[p, :len(kc)], wa), (th, p, nm[p], wn)
            assert wn > 3000
    # the local map against a resident frame (mode 3)
    rng = np.random.default_rng(1212)
    kc, dc = host[2]
    start, idx = oracle.grid_build(gp, kc)
    for th, nq, with_occ in ((1.0, 9000, False), (3.0, 15000, True)):
        uvr, ql, qd, qv, qo = local_map_queries(rng, host[:2], sf, th, nq)
        occ = (rng.random(len(kc)) < 0.3).astype(np.uint8) if with_occ else np.zeros(len(kc), np.uint8)
        fs.track_local_points(2, uvr, ql, qd, qv, qo, occ if with_occ else None)
        assign, nm = fs.results()
        wa, _, wn = oracle.search_by_projection(3, 0.8, True, 100, uvr, ql, qd, None, qv, qo, gp, kc, start, idx, dc, occ, np.full(len(kc), -1, np.int32))
        assert nm[0] == wn and np.array_equal(assign[0, :len(kc)], wa), (th, nq, nm[0], wn)
        assert wn > 1000
    # LastFrame's features with the caller's projections (modes 4 and 5)
    kl, dl = host[1]
    for mode, th, ori, thd in ((4, 15.0, True, 100), (5, 10.0, True, 64)):
        u = kl["x"] * np.float32(1.002) + np.float32(1.5)
        v = kl["y"] * np.float32(1.002) - np.float32(0.8)
        uvr = np.stack([u, v, np.float32(th) * sf[kl["octave"]]], axis=1).astype(np.float32)
        lvl = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
        qv = ((rng.random(len(kl)) < 0.85) & (u >= 0) & (u <= w) & (v >= 0) & (v <= h)).astype(np.uint8)
        qo = (rng.random(len(kl)) < 0.9).astype(np.uint8)
        fs.track_projected(2, 1, uvr, lvl, qv, qo, None, th_dist=thd, nnratio=0.9, check_ori=ori, mode=mode)
        assign, nm = fs.results()
        wa, _, wn = oracle.search_by_projection(mode, 0.9, ori, thd, uvr, lvl, dl, kl["angle"], qv, qo, gp, kc, start, idx, dc,
                                                np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32))
        assert nm[0] == wn and np.array_equal(assign[0, :len(kc)], wa), (mode, nm[0], wn)
        assert wn > 3000
    fs.close(); m.close(); gex.close()


def test_device_frame_and_bow_of_large_frames(gpu, oracle):
    """orbm_frame_create on 12 000 device-resident features (undistorted), a projection search against it, and
    ORBVocabulary::transform / Frame::ComputeBoW + SearchByBoW on frames of that size (the sort's keys in memory)"""
    from orbslamm_amd import ORBmatcher, ORBVocabulary, make_grid
    from vocab_cases import make_vocab
    w, h = 1920, 1080
    gex, host = _big_extraction(oracle, nframes=2, stream=78)
    dk, dd, _, cap = gex.device_results()
    D = [0.05, -0.02, 0.0005, -0.0003, 0.004]
    rng = np.random.default_rng(88)
    keys, desc = host[1]
    nt = len(keys)
    want_un = oracle.undistort_keypoints(keys, K_BIG, D)
    x0, x1 = float(np.floor(want_un["x"].min())) - 1, float(np.ceil(want_un["x"].max())) + 1
    y0, y1 = float(np.floor(want_un["y"].min())) - 1, float(np.ceil(want_un["y"].max())) + 1
    g = make_grid(x0, y0, x1, y1)
    gp = oracle.make_grid_params(x0, y0, x1, y1)
    m = ORBmatcher(0.9, True, device=0)
    frame = m.frame_from_device(dk + 1 * cap * 28, dd + 1 * cap * 32, nt, K_BIG, D, g)
    assert m.frame_keys_un(frame).tobytes() == want_un.tobytes()
    start, idx = oracle.grid_build(gp, want_un)
    nq = 6000
    src = rng.integers(0, nt, nq)
    qd = noisy_copies(rng, desc[src], 10)
    uvr = np.stack([want_un["x"][src] + rng.normal(0, 2, nq), want_un["y"][src] + rng.normal(0, 2, nq),
                    15.0 * np.float32(1.2) ** want_un["octave"][src]], axis=1).astype(np.float32)
    lvl = np.stack([want_un["octave"][src] - 1, want_un["octave"][src] + 1], axis=1).astype(np.int8)
    qa = want_un["angle"][src].astype(np.float32)
    occ = (rng.random(nt) < 0.1).astype(np.uint8)
    a0 = np.full(nt, -1, np.int32)
    ga, gocc, gn = m.SearchByProjectionFrame(4, 100, uvr, lvl, qd, qa, None, None, frame, occ, a0)
    wa, wocc, wn = oracle.search_by_projection(4, 0.9, True, 100, uvr, lvl, qd, qa, None, None, gp, want_un, start, idx, desc, occ, a0)
    assert gn == wn and np.array_equal(ga, wa) and np.array_equal(gocc, wocc) and wn > 2000
    m.frame_destroy(frame)

    # the vocabulary transform on more than 8 192 descriptors, flat and on a frame set
    voc = make_vocab(rng, 10, 4)
    for scoring, weighting in ((0, 0), (1, 1)):
        G = ORBVocabulary(10, 4, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
        O = oracle.Vocabulary(10, 4, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        for dsc in (desc, rng.integers(0, 256, size=(20001, 32), dtype=np.uint8)):
            (gi, gv), (gn_, gs, gx) = G.transform(dsc, 2)
            (oi, ov), (on, os_, ox) = O.transform(dsc, 2)
            assert np.array_equal(gi, oi) and gv.tobytes() == ov.tobytes()
            assert np.array_equal(gn_, on) and np.array_equal(gs, os_) and np.array_equal(gx, ox)
        if scoring == 0:
            gb = make_grid(0.0, 0.0, float(w), float(h))
            fs = m.frame_set(3, gex.max_keypoints, K_BIG, [0, 0, 0, 0, 0], gb, [0.0, float(w), 0.0, float(h)], np.array(gex.GetScaleFactors(), np.float32))
            fs.build_from_extractor(1, gex)          # frames 0, 1 -> slots 1, 2
            fs.compute_bow(G, 1, 2, 2)
            fvs = []
            for f in range(2):
                (owid, owval), ofv = O.transform(host[f][1], 2)
                wid, wval = fs.bow_vector(1 + f)
                assert np.array_equal(wid, owid) and wval.tobytes() == owval.tobytes()
                fvs.append(ofv)
            fs.search_by_bow([1], [2], nnratio=0.7, check_ori=True)
            match, nm = fs.bow_results()
            (kq, dq), (kt, dt) = host[0], host[1]
            want, wn2 = oracle.search_by_bow(dq, kq["angle"], None, fvs[0], dt, kt["angle"], None, fvs[1], 0.7, True, True)
            assert nm[0] == wn2 and np.array_equal(match[0, :len(kt)], want) and wn2 > 500
            fs.close()
        G.close()
    m.close(); gex.close()
