"""GPU parity of ORBVocabulary::transform (through the C ABI) against the oracle."""
import numpy as np
import pytest

from vocab_cases import make_vocab, write_text

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 1), (5, 0), (0, 2), (5, 3), (3, 0)])
def test_transform_parity(gpu, oracle, scoring, weighting):
    from orbslamm_amd import ORBVocabulary
    rng = np.random.default_rng(60 + scoring * 4 + weighting)
    for k, L, n in ((10, 3, 2000), (6, 4, 333), (3, 6, 1), (10, 4, 4097)):
        voc = make_vocab(rng, k, L)
        G = ORBVocabulary(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
        O = oracle.Vocabulary(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
        desc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        for levelsup in (4, 0, L + 2):
            (gi, gv), (gn, gs, gx) = G.transform(desc, levelsup)
            (oi, ov), (on, os_, ox) = O.transform(desc, levelsup)
            assert np.array_equal(gi, oi)
            assert gv.tobytes() == ov.tobytes()   # double values bit-for-bit
            assert np.array_equal(gn, on) and np.array_equal(gs, os_) and np.array_equal(gx, ox)
        assert len(oi) > 0


def test_text_loader_and_bow_search(gpu, oracle, tmp_path):
    """ORBvoc.txt-format file -> transform -> FeatureVector drives SearchByBoW"""
    from orbslamm_amd import ORBmatcher, ORBVocabulary
    from matcher_cases import noisy_copies
    rng = np.random.default_rng(77)
    voc = make_vocab(rng, 10, 3, ragged=False)
    path = tmp_path / "voc.txt"
    write_text(path, voc, 0, 0)
    G = ORBVocabulary.loadFromTextFile(path)
    O = oracle.Vocabulary(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    base = rng.integers(0, 256, size=(800, 32), dtype=np.uint8)
    d1, d2 = noisy_copies(rng, base, 6), noisy_copies(rng, base, 6)[rng.permutation(800)]
    (w1, v1), fv1 = G.transform(d1, 2)
    (w2, v2), fv2 = G.transform(d2, 2)
    (ow1, ov1), ofv1 = O.transform(d1, 2)
    assert np.array_equal(w1, ow1) and v1.tobytes() == ov1.tobytes() and all(np.array_equal(a, b) for a, b in zip(fv1, ofv1))
    ang = np.zeros(800, np.float32)
    m = ORBmatcher(0.8, False, device=0)
    got, n = m.SearchByBoW(d1, ang, None, fv1, d2, ang, None, fv2, True)
    want, nw = oracle.search_by_bow(d1, ang, None, fv1, d2, ang, None, fv2, 0.8, False, True)
    assert n == nw and np.array_equal(got, want) and n > 300


def test_bad_vocabulary_is_rejected(gpu):
    from orbslamm_amd import ORBVocabulary, OrbError
    with pytest.raises(OrbError):
        ORBVocabulary(10, 3, 0, 0, np.array([5], np.int32), np.array([1], np.uint8), np.zeros((1, 32), np.uint8), np.ones(1), device=0)
    with pytest.raises(OrbError):
        ORBVocabulary(99, 3, 0, 0, np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 32), np.uint8), np.zeros(0), device=0)


def test_tracking_front_end_stays_in_hbm(gpu, oracle):
    """SURVEY 8(f).2+3 together: frames extracted on the GPU never leave it -- the matcher builds their Frame state
    from the device pointers, Frame::ComputeBoW runs on the device-resident descriptors, and SearchByBoW (the
    TrackReferenceKeyFrame search) runs between the two device frames.  BowVector, and matches against the oracle
    fed with the downloaded arrays."""
    from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth
    w, h, nf = 640, 480, 1000
    rng = np.random.default_rng(91)
    voc = make_vocab(rng, 10, 4)
    G = ORBVocabulary(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
    O = oracle.Vocabulary(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    fr = synth.make_frames(w, h, 2, stream=4)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, device=0)
    gex.extract_batch_device(*gex.upload_frames(fr))
    gex.sync()
    dk, dd, _, cap = gex.device_results()
    host = [gex.download(f) for f in range(2)]
    g = make_grid(0.0, 0.0, float(w), float(h))
    K, D0 = [517.3, 516.5, 318.6, 255.3], [0, 0, 0, 0, 0]
    for ratio, ori, by_train in ((0.7, True, True), (0.75, True, False), (0.9, False, True)):
        m = ORBmatcher(ratio, ori, device=0)
        frames, fvs = [], []
        for f in range(2):
            keys, desc = host[f]
            F = m.frame_from_device(dk + f * cap * 28, dd + f * cap * 32, len(keys), K, D0, g)
            wid, wval = m.frame_compute_bow(F, G, 4)
            (owid, owval), ofv = O.transform(desc, 4)
            assert np.array_equal(wid, owid) and wval.tobytes() == owval.tobytes()
            frames.append(F); fvs.append(ofv)
        (k0, d0), (k1, d1) = host
        qv = (rng.uniform(size=len(k0)) < 0.9).astype(np.uint8)
        tv = None if by_train else (rng.uniform(size=len(k1)) < 0.9).astype(np.uint8)
        got, n = m.SearchByBoWFrames(frames[0], qv, frames[1], tv, by_train)
        want, nw = oracle.search_by_bow(d0, k0["angle"], qv, fvs[0], d1, k1["angle"], tv, fvs[1], ratio, ori, by_train)
        assert n == nw and np.array_equal(got, want) and nw > 50
        # monocular initialisation search between the same two device frames (ORBmatcher.cc:407)
        q_xy = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32)
        gp = oracle.make_grid_params(0.0, 0.0, float(w), float(h))
        start, idx = oracle.grid_build(gp, k1)
        gi, gni = m.SearchForInitializationFrames(q_xy, 100.0, frames[0], frames[1])
        wi, wni = oracle.search_for_initialization(q_xy, 100.0, k0, d0, gp, k1, start, idx, d1, ratio, ori)
        assert gni == wni and np.array_equal(gi, wi) and wni > 30
        # SearchForTriangulation between the two device frames (ORBmatcher.cc:659): descriptors, keys and FeatureVectors
        # stay in HBM, only the "already has a MapPoint" flags and the fundamental matrix come from the host.
        # F12 of a pure sideways translation: epipolar lines are the rows, the epipole lies at infinity
        sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
        F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float32)
        s1 = (rng.uniform(size=len(k0)) < 0.3).astype(np.uint8)
        s2 = (rng.uniform(size=len(k1)) < 0.3).astype(np.uint8)
        gt, gnt = m.SearchForTriangulationFrames(frames[0], s1, frames[1], s2, F12, 1.0e6, 240.0, sf, sf * sf)
        wt, wnt = oracle.search_for_triangulation(k0, d0, s1, fvs[0], k1, d1, s2, fvs[1], F12, 1.0e6, 240.0, sf, sf * sf, False, ori)
        assert gnt == wnt and np.array_equal(gt, wt) and wnt > 20
        # the windowed best search of Fuse / SearchBySim3 with the device frame as train side
        nq = 600
        pick = rng.integers(0, len(k1), nq)
        uvr = np.stack([k1["x"][pick] + rng.uniform(-2, 2, nq), k1["y"][pick] + rng.uniform(-2, 2, nq),
                        3.0 * sf[np.clip(k1["octave"][pick], 0, 7)]], axis=1).astype(np.float32)
        pred = np.clip(k1["octave"][pick] + rng.integers(0, 2, nq), 0, 7).astype(np.int8)
        qd = d1[pick].copy()
        flip = rng.integers(0, 32, nq)
        qd[np.arange(nq), flip] ^= np.uint8(5)
        inv = (1.0 / (sf * sf)).astype(np.float32)
        for chi2 in (False, True):
            bi, bd = m.window_best_frame(uvr, pred, qd, None, frames[1], inv, chi2)
            wi_, wd_ = oracle.window_best(uvr, pred, qd, None, gp, k1, start, idx, d1, inv, chi2)
            assert np.array_equal(bi, wi_) and np.array_equal(bd, wd_) and (wi_ >= 0).sum() > 300
        for F in frames:
            m.frame_destroy(F)
