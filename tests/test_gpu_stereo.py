"""Frame::ComputeStereoMatches (src/Frame.cc:466-638) on the GPU against the oracle's line-by-line restatement:
mvuRight and mvDepth bit for bit (binary32), on rectified synthetic pairs whose disparity varies over the image."""
import numpy as np
import pytest

from conftest import frames_for

pytestmark = pytest.mark.gpu


def _pair(w, h, stream, seed):
    """right(x) = left(x + d): disparity d by horizontal band, fresh sensor noise on the right image"""
    left = frames_for(w, h, 1, stream=stream)[0]
    rng = np.random.default_rng(seed)
    right = np.empty_like(left)
    bands = [(0, h // 3, 6), (h // 3, 2 * h // 3, 14), (2 * h // 3, h, 27)]
    for y0, y1, d in bands:
        right[y0:y1] = np.roll(left[y0:y1], -d, axis=1)
    noise = rng.integers(-3, 4, size=right.shape)
    return left, np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def _levels(oracle, oex, pyr, w, h):
    out, o = [], 0
    for l in range(8):
        lw, lh = oex.level_size(w, h, l)
        out.append(pyr[o:o + lw * lh].reshape(lh, lw))
        o += lw * lh
    return out


@pytest.mark.parametrize("w,h,nf,mb,mbf", [(640, 480, 1000, 0.08, 40.0), (1241, 376, 2000, 0.54, 386.1), (752, 480, 1200, 0.11, 47.9)])
def test_compute_stereo_matches_parity(gpu, oracle, w, h, nf, mb, mbf):
    from orbslamm_amd import ORBextractor
    left, right = _pair(w, h, stream=3, seed=w)
    exL = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    exR = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    kL, dL = exL(left)      # Frame::Frame(stereo): the two extractions run on two threads in the reference (Frame.cc:78-81)
    kR, dR = exR(right)
    oex = oracle.Extractor(nf, 1.2, 8, 20, 7)
    oL, oR = oex(left, want_pyramid=True), oex(right, want_pyramid=True)
    assert kL.tobytes() == oL["kps"].tobytes() and kR.tobytes() == oR["kps"].tobytes()
    sf = oex.scale_factors()
    want_u, want_d, accepted = oracle.compute_stereo_matches(oL["kps"], oL["desc"], oR["kps"], oR["desc"], _levels(oracle, oex, oL["pyramid"], w, h),
                                                             _levels(oracle, oex, oR["pyramid"], w, h), sf, (1.0 / sf).astype(np.float32), mb, mbf)
    # the inverse scale factors are the extractor's own table (1.0f / mvScaleFactor[i], ORBextractor.cc:424-428)
    assert np.array_equal((np.float32(1.0) / sf).astype(np.float32), exL.GetInverseScaleFactors())
    got_u, got_d = exL.compute_stereo_matches(exR, mb, mbf)
    assert got_u.tobytes() == want_u.tobytes()
    assert got_d.tobytes() == want_d.tobytes()
    matched = got_u >= 0
    assert accepted > 100 and matched.sum() > 80            # the scenario is meaningful
    disp = kL["x"][matched] - got_u[matched]
    band = np.select([kL["y"][matched] < h // 3, kL["y"][matched] < 2 * h // 3], [6, 14], 27)
    assert np.mean(np.abs(disp - band) < 1.0) > 0.8          # sub-pixel disparities land on the true ones
    assert np.allclose(got_d[matched], np.float32(mbf) / np.maximum(disp, 0.01), rtol=1e-5)


def test_stereo_argument_errors(gpu):
    from orbslamm_amd import ORBextractor, OrbError
    a = ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=1, device=0)
    b = ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1, device=0)
    fa = frames_for(320, 240, 1)[0]
    fb = frames_for(640, 480, 1)[0]
    a(fa); b(fb)
    with pytest.raises(OrbError):
        a.compute_stereo_matches(b, 0.1, 40.0)     # different shapes
    b(fa)
    with pytest.raises(OrbError):
        a.compute_stereo_matches(b, 0.0, 40.0)     # no baseline
    # the same image left and right: every window distance is 0, so the median is 0 and the filter `dist < 2.1 * median`
    # (:628-637) throws every match away -- the reference's behaviour, literally
    u, d = a.compute_stereo_matches(b, 0.1, 40.0)
    assert len(u) > 100 and np.all(u == -1) and np.all(d == -1)


def test_compute_stereo_from_rgbd_parity(gpu, oracle):
    """Frame::ComputeStereoFromRGBD (Frame.cc:641-663, the RGB-D constructor's tail): depth under the DISTORTED keypoint
    (float -> int by truncation), mvuRight from the UNDISTORTED x; holes (d <= 0, NaN) and keypoints outside the depth
    image give -1 / -1.  Byte for byte against the oracle."""
    from orbslamm_amd import ORBextractor, ORBmatcher
    w, h, nf = 640, 480, 1000
    K = [517.306408, 516.469215, 318.643040, 255.313989]
    D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    fr = frames_for(w, h, 1, stream=12)[0]
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1, device=0)
    keys, _ = gex(fr)
    m = ORBmatcher(0.9, True, device=0)
    keys_un = m.UndistortKeyPoints(keys, K, D)
    rng = np.random.default_rng(641)
    depth = rng.uniform(0.3, 8.0, size=(h, w)).astype(np.float32)
    depth[rng.random((h, w)) < 0.2] = 0.0          # holes of the sensor
    depth[rng.random((h, w)) < 0.02] = -1.0
    depth[rng.random((h, w)) < 0.01] = np.nan
    for dimg, kk in ((depth, keys), (depth[:300, :500], keys)):      # the second: a depth image smaller than the frame
        want_u, want_d = oracle.stereo_from_rgbd(kk, keys_un, dimg, 40.0)
        got_u, got_d = m.ComputeStereoFromRGBD(kk, keys_un, dimg, 40.0)
        assert got_u.tobytes() == want_u.tobytes() and got_d.tobytes() == want_d.tobytes()
        assert (got_d > 0).sum() > 200 and (got_d == -1).sum() > 50
    u0, d0 = m.ComputeStereoFromRGBD(keys[:0], keys_un[:0], depth, 40.0)
    assert len(u0) == 0 and len(d0) == 0
    m.close(); gex.close()
