"""GPU parity on NATURAL images (tests/golden/natural.npz: skimage.data photographs and textures at the benchmark
shapes, the Middlebury motorcycle stereo pair): extractor, brute-force match, frame-to-frame projection search and
Frame::ComputeStereoMatches, each bit for bit against the CPU oracle on the same bytes."""
import numpy as np
import pytest

from natural_cases import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(gpu, oracle):
    """every frame through both extractors once"""
    from orbslamm_amd import ORBextractor
    frames, digests = load()
    out = {}
    exs = {}
    for name, (img, nf) in frames.items():
        h, w = img.shape
        key = (w, h, nf)
        if key not in exs:
            exs[key] = (ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0), oracle.Extractor(nf, 1.2, 8, 20, 7))
        gex, oex = exs[key]
        out[name] = dict(img=img, nf=nf, g=gex(img), o=oex(img))
    return out


def test_extractor_bit_exact_on_natural_images(nat):
    total = 0
    for name, r in nat.items():
        gk, gd = r["g"]
        assert len(gk) == len(r["o"]["kps"]) > 300, name
        assert gk.tobytes() == r["o"]["kps"].tobytes(), name
        assert gd.tobytes() == r["o"]["desc"].tobytes(), name
        total += len(gk)
    assert total > 15000
    # all eight levels are populated on photographs too
    assert set(np.unique(nat["c3_mosaic"]["g"][0]["octave"])) == set(range(8))


@pytest.mark.parametrize("a,b", [("c3_pan1", "c3_pan0"), ("c3_pan2", "c3_pan1"), ("c3_hubble", "c3_mosaic"), ("c2_camera", "c2_astronaut"),
                                 ("c2_grass", "c2_brick"), ("stereo_r", "stereo_l")])
def test_bruteforce_match_on_natural_images(nat, oracle, a, b):
    from orbslamm_amd import ORBmatcher
    (qk, qd), (tk, td) = nat[a]["g"], nat[b]["g"]
    m = ORBmatcher(0.7, True, device=0)
    got, gn = m.match_bruteforce(qd, qk["angle"], td, tk["angle"])
    want, wn = oracle.match_bruteforce(qd, qk["angle"], td, tk["angle"], 0.7, 50, True)
    assert gn == wn and np.array_equal(got, want)
    if a.startswith("c3_pan") or a.startswith("stereo"):
        assert wn > 150   # the same scene seen twice


def test_tracking_search_on_a_natural_sequence(nat, oracle):
    """SearchByProjection(Cur, Last) over the panning sequence: host-array entry against the oracle (th 15 and 30)"""
    from orbslamm_amd import ORBmatcher, make_grid
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sf = np.array(oracle.Extractor(2000, 1.2, 8, 20, 7).scale_factors(), np.float32)
    g = make_grid(0.0, 0.0, 1241.0, 376.0)
    gp = oracle.make_grid_params(0.0, 0.0, 1241.0, 376.0)
    for cur, last in (("c3_pan1", "c3_pan0"), ("c3_pan2", "c3_pan1")):
        (kc, dc), (kl, dl) = nat[cur]["g"], nat[last]["g"]
        start, idx = oracle.grid_build(gp, kc)
        for th in (15.0, 30.0):
            uvr = np.stack([kl["x"], kl["y"], (np.float32(th) * sf[kl["octave"]]).astype(np.float32)], axis=1).astype(np.float32)
            lvl = np.stack([kl["octave"] - 1, kl["octave"] + 1], axis=1).astype(np.int8)
            m = ORBmatcher(0.9, True, device=0)
            occ0, a0 = np.zeros(len(kc), np.uint8), np.full(len(kc), -1, np.int32)
            ga, gocc, gn = m.SearchByProjection(4, 100, uvr, lvl, dl, kl["angle"], None, None, g, kc, dc, occ0, a0)
            wa, wocc, wn = oracle.search_by_projection(4, 0.9, True, 100, uvr, lvl, dl, kl["angle"], None, None, gp, kc, start, idx, dc, occ0, a0)
            assert gn == wn and wn > 300 and np.array_equal(ga, wa) and np.array_equal(gocc, wocc)


def test_compute_stereo_matches_on_the_motorcycle_pair(gpu, oracle):
    from orbslamm_amd import ORBextractor
    frames, _ = load()
    left, right = frames["stereo_l"][0], frames["stereo_r"][0]
    h, w = left.shape
    nf, mb, mbf = 1200, 0.193, 193.0 * 3.98   # baseline 193 mm; focal length ~3980 px at full resolution / 4
    exL = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    exR = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    kL, dL = exL(left)
    kR, dR = exR(right)
    oex = oracle.Extractor(nf, 1.2, 8, 20, 7)
    oL, oR = oex(left, want_pyramid=True), oex(right, want_pyramid=True)
    assert kL.tobytes() == oL["kps"].tobytes() and kR.tobytes() == oR["kps"].tobytes()

    def levels(pyr):
        out, o = [], 0
        for l in range(8):
            lw, lh = oex.level_size(w, h, l)
            out.append(pyr[o:o + lw * lh].reshape(lh, lw))
            o += lw * lh
        return out

    sf = oex.scale_factors()
    want_u, want_d, accepted = oracle.compute_stereo_matches(oL["kps"], oL["desc"], oR["kps"], oR["desc"], levels(oL["pyramid"]), levels(oR["pyramid"]),
                                                             sf, (1.0 / sf).astype(np.float32), mb, mbf)
    got_u, got_d = exL.compute_stereo_matches(exR, mb, mbf)
    assert got_u.tobytes() == want_u.tobytes() and got_d.tobytes() == want_d.tobytes()
    assert (got_u >= 0).sum() > 100   # a real stereo pair: plenty of matches survive the SAD refinement and the median filter
