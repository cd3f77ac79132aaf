"""GPU parity of the matcher entry points (through the C ABI) against the oracle."""
import numpy as np
import pytest

from conftest import random_descriptors
from matcher_cases import make_bow_case, make_proj_case, noisy_copies

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm(gpu):
    from orbslamm_amd import ORBmatcher
    return ORBmatcher(0.7, True, device=0)


def test_descriptor_distance(gm, oracle):
    rng = np.random.default_rng(21)
    q, t = random_descriptors(rng, 70), random_descriptors(rng, 333)
    D = gm.distance_matrix(q, t)
    want = (np.unpackbits(q, axis=1)[:, None, :] != np.unpackbits(t, axis=1)[None, :, :]).sum(-1)
    assert np.array_equal(D, want)
    assert gm.DescriptorDistance(q[0], q[0]) == 0
    assert gm.DescriptorDistance(q[0], ~q[0]) == 256
    assert gm.DescriptorDistance(q[1], t[5]) == oracle.descriptor_distance(q[1], t[5])


@pytest.mark.parametrize("nq,nt,ori", [(2000, 2000, True), (1, 1, True), (257, 1023, False), (5, 0, True), (900, 40, True)])
def test_bruteforce_parity(gm, oracle, nq, nt, ori):
    from orbslamm_amd import ORBmatcher
    rng = np.random.default_rng(nq * 7 + nt)
    base = random_descriptors(rng, max(nq, nt, 1))
    q = noisy_copies(rng, base[:nq], 12)
    t = base[:nt][rng.permutation(nt)] if nt else np.zeros((0, 32), np.uint8)
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    m = ORBmatcher(0.7, ori, device=0)
    got, n = m.match_bruteforce(q, qa, t, ta)
    want, nw = oracle.match_bruteforce(q, qa, t, ta, 0.7, 50, ori)
    assert n == nw and np.array_equal(got, want)


def test_bruteforce_ties_first_wins(gm, oracle):
    # duplicated train descriptors: the lowest index must win, second == best blocks the ratio test
    rng = np.random.default_rng(22)
    t = random_descriptors(rng, 64)
    t[40] = t[3]
    q = t[[3, 10]].copy()
    q[1, 0] ^= 1
    ang = np.zeros(64, np.float32)
    got, n = gm.match_bruteforce(q, ang[:2], t, ang)
    want, nw = oracle.match_bruteforce(q, ang[:2], t, ang, 0.7, 50, True)
    assert np.array_equal(got, want) and n == nw
    assert got[0] == -1 and got[1] == 10  # 0 < 0.7*0 fails for the duplicate


@pytest.mark.parametrize("nq,nt", [(31, 31), (32, 32), (33, 33), (63, 65), (64, 64), (255, 97), (256, 256), (257, 96), (513, 31)])
def test_bruteforce_tile_boundaries(gm, oracle, nq, nt):
    """the matrix-core scan works on blocks of 32 trains / 64 queries per wave: sizes around those edges,
    with duplicates planted in different tiles (lowest index wins, equal second blocks the ratio test)"""
    rng = np.random.default_rng(1000 * nq + nt)
    base = random_descriptors(rng, max(nq, nt))
    t = base[:nt].copy()
    q = noisy_copies(rng, base[rng.integers(0, nt, nq)], 9)
    if nt > 40:
        t[nt - 1] = t[2]          # duplicate in the last (partial) tile
        t[37] = t[5]              # duplicate in the next tile
        q[0] = t[2]; q[1] = t[5]; q[1, 3] ^= 4
    q[-1] = ~t[0]                 # distance 256 to train 0
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    for ori in (True, False):
        from orbslamm_amd import ORBmatcher
        got, n = ORBmatcher(0.7, ori, device=0).match_bruteforce(q, qa, t, ta)
        want, nw = oracle.match_bruteforce(q, qa, t, ta, 0.7, 50, ori)
        assert n == nw and np.array_equal(got, want)


@pytest.mark.parametrize("nq,nt", [(300, 2048), (70, 2049), (257, 4097), (40, 5000), (33, 65535)])
def test_bruteforce_more_than_sixty_four_tiles(gm, oracle, nq, nt):
    """the scan's running keys carry the tile as an offset that lasts 64 tiles; beyond that they are decoded and merged
    epoch by epoch: duplicates planted across the epoch edges (the lowest index wins, an equal second blocks the ratio
    test), best and second in different epochs"""
    rng = np.random.default_rng(nq + nt)
    t = random_descriptors(rng, nt)
    q = noisy_copies(rng, t[rng.integers(0, nt, nq)], 9)
    t[nt - 1] = t[7]                 # first and last epoch
    q[0] = t[7]
    if nt > 2100:
        t[2050] = t[2040]            # either side of the first epoch edge
        q[1] = t[2040]; q[1, 5] ^= 16
        q[2] = t[nt - 3]; q[2, 0] ^= 3    # best in the last epoch, second elsewhere
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    from orbslamm_amd import ORBmatcher
    for ori in (True, False):
        got, n = ORBmatcher(0.7, ori, device=0).match_bruteforce(q, qa, t, ta)
        want, nw = oracle.match_bruteforce(q, qa, t, ta, 0.7, 50, ori)
        assert n == nw and np.array_equal(got, want)
    assert got[0] == -1 and (nt <= 2100 or (got[1] == -1 and got[2] == nt - 3))  # (without the rotation check)


def test_bruteforce_wide_train_fallback(gm, oracle):
    """more than 65535 train features: the popcount scan (k_match_best2) takes over"""
    rng = np.random.default_rng(77)
    nt, nq = 66000, 24
    t = random_descriptors(rng, nt)
    q = noisy_copies(rng, t[rng.integers(0, nt, nq)], 10)
    q[3] = t[65990]
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    got, n = gm.match_bruteforce(q, qa, t, ta)
    want, nw = oracle.match_bruteforce(q, qa, t, ta, 0.7, 50, True)
    assert n == nw and np.array_equal(got, want) and got[3] == 65990


@pytest.mark.parametrize("by_train", [True, False])
@pytest.mark.parametrize("nq,nt,nnodes,ratio", [(400, 450, 23, 0.75), (2000, 2000, 97, 0.7), (300, 300, 1, 0.9), (50, 700, 5, 0.6)])
def test_search_by_bow_parity(gpu, oracle, by_train, nq, nt, nnodes, ratio):
    from orbslamm_amd import ORBmatcher
    rng = np.random.default_rng(nq + nt + nnodes)
    c = make_bow_case(rng, nq, nt, nnodes)
    for ori in (True, False):
        m = ORBmatcher(ratio, ori, device=0)
        tv = None if by_train else c["tv"]
        got, n = m.SearchByBoW(c["qd"], c["qa"], c["qv"], c["qfv"], c["td"], c["ta"], tv, c["tfv"], by_train)
        want, nw = oracle.search_by_bow(c["qd"], c["qa"], c["qv"], c["qfv"], c["td"], c["ta"], tv, c["tfv"], ratio, ori, by_train)
        assert n == nw and np.array_equal(got, want)
        assert nw > 0


def test_grid_features_in_area_order(gm, oracle):
    from orbslamm_amd import make_grid
    rng = np.random.default_rng(23)
    n = 2000
    keys = np.zeros(n, dtype=oracle.KP_DTYPE)
    keys["x"] = rng.uniform(-5, 1246, n).astype(np.float32)  # some fall outside the grid
    keys["y"] = rng.uniform(-5, 381, n).astype(np.float32)
    keys["octave"] = rng.integers(0, 8, n)
    gp = oracle.make_grid_params(0.0, 0.0, 1241.0, 376.0)
    g = make_grid(0.0, 0.0, 1241.0, 376.0)
    start, idx = oracle.grid_build(gp, keys)
    for _ in range(40):
        x, y, r = rng.uniform(-30, 1270), rng.uniform(-30, 400), rng.uniform(2, 120)
        lo, hi = [(-1, -1), (0, 2), (3, 5), (2, -1), (0, -1), (7, 8)][rng.integers(0, 6)]
        want = oracle.features_in_area(gp, keys, start, idx, x, y, r, lo, hi)
        got = gm.GetFeaturesInArea(g, keys, x, y, r, lo, hi)
        assert np.array_equal(got, want)  # same indices in the same (cell-major) order


@pytest.mark.parametrize("mode,th,ratio", [(3, 100, 0.8), (4, 100, 0.9), (5, 100, 0.9), (5, 64, 0.9), (6, 50, 0.75)])
def test_search_by_projection_parity(gpu, oracle, mode, th, ratio):
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(100 + mode + th)
    for nq, nt in ((300, 900), (1500, 2000), (40, 3000)):
        c = make_proj_case(rng, nq, nt)
        g = make_grid(0.0, 0.0, c["w"], c["h"])
        for ori in (True, False):
            m = ORBmatcher(ratio, ori, device=0)
            a0 = np.full(nt, -1, np.int32)
            ga, gocc, gn = m.SearchByProjection(mode, th, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], g,
                                                c["tk"], c["td"], c["occ"], a0)
            wa, wocc, wn = oracle.search_by_projection(mode, ratio, ori, th, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"],
                                                       c["qo"], c["gp"], c["tk"], c["start"], c["idx"], c["td"], c["occ"], a0)
            assert gn == wn and np.array_equal(ga, wa) and np.array_equal(gocc, wocc)
        assert wn > 0


def test_projection_prepare_is_a_hint_not_a_contract(gpu, oracle):
    """orbm_projection_prepare (the drop-in members upload the train frame under their MapPoint walk): a search that follows
    with the SAME arrays finds the frame resident, one that follows with OTHER arrays (or another grid, or after a second
    search) uploads its own -- the results are the oracle's either way, in the LDS form and in the memory form"""
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(4711)
    m = ORBmatcher(0.9, True, device=0)
    for nq, nt in ((1500, 2000), (700, 9000)):
        c, d = make_proj_case(rng, nq, nt), make_proj_case(rng, nq, nt)
        g = make_grid(0.0, 0.0, c["w"], c["h"])
        g2 = make_grid(1.0, 0.0, c["w"], c["h"])
        a0 = np.full(nt, -1, np.int32)

        def want(cc):
            return oracle.search_by_projection(4, 0.9, True, 100, cc["uvr"], cc["lvl"], cc["qd"], cc["qa"], cc["qv"], cc["qo"], cc["gp"],
                                               cc["tk"], cc["start"], cc["idx"], cc["td"], cc["occ"], a0)

        def got(cc):
            return m.SearchByProjection(4, 100, cc["uvr"], cc["lvl"], cc["qd"], cc["qa"], cc["qv"], cc["qo"], g, cc["tk"], cc["td"], cc["occ"], a0)

        def same(x, y):
            return x[2] == y[2] and np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        wc, wd = want(c), want(d)
        m.ProjectionPrepare(g, c["tk"], c["td"]); assert same(got(c), wc)      # prepared and used
        assert same(got(c), wc)                                                # the hint is spent: own upload, same result
        m.ProjectionPrepare(g, c["tk"], c["td"]); assert same(got(d), wd)      # prepared with other arrays: ignored
        m.ProjectionPrepare(g2, c["tk"], c["td"]); assert same(got(c), wc)     # prepared on another grid: ignored
        m.ProjectionPrepare(g, c["tk"], c["td"]); m.ProjectionPrepare(g, d["tk"], d["td"]); assert same(got(d), wd)   # the later prepare counts
        assert wc[2] > 100 and wd[2] > 100
    m.close()


def test_projection_overwrite_semantics(gpu, oracle):
    """mode 4: a MapPoint without observations does not block the feature, a later query
    takes it again and both matches are counted (ORBmatcher.cc:1405-1407, 1430-1431)"""
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(31)
    tk = np.zeros(4, dtype=oracle.KP_DTYPE)
    tk["x"], tk["y"], tk["octave"] = [100, 300, 500, 700], [100, 100, 100, 100], [0, 0, 0, 0]
    td = random_descriptors(rng, 4)
    qd = np.stack([td[1], td[1], td[1]])
    uvr = np.array([[300, 100, 15]] * 3, np.float32)
    lvl = np.array([[-1, 1]] * 3, np.int8)
    g = make_grid(0.0, 0.0, 1241.0, 376.0)
    gp = oracle.make_grid_params(0.0, 0.0, 1241.0, 376.0)
    start, idx = oracle.grid_build(gp, tk)
    for obs, want_n in (([0, 0, 1], 3), ([1, 0, 0], 1)):
        m = ORBmatcher(0.9, False, device=0)
        a0 = np.full(4, -1, np.int32)
        occ0 = np.zeros(4, np.uint8)
        ga, gocc, gn = m.SearchByProjection(4, 100, uvr, lvl, qd, np.zeros(3, np.float32), None, np.array(obs, np.uint8), g, tk, td, occ0, a0)
        wa, wocc, wn = oracle.search_by_projection(4, 0.9, False, 100, uvr, lvl, qd, np.zeros(3, np.float32), None,
                                                   np.array(obs, np.uint8), gp, tk, start, idx, td, occ0, a0)
        assert gn == wn == want_n and np.array_equal(ga, wa) and np.array_equal(gocc, wocc)


@pytest.mark.parametrize("chi2", [False, True])
def test_window_best_parity(gpu, oracle, chi2):
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(200 + chi2)
    m = ORBmatcher(0.8, True, device=0)
    inv = (1.0 / (np.float32(1.2) ** np.arange(8)) ** 2).astype(np.float32)
    for nq, nt, stereo in ((300, 900, False), (2000, 2000, False), (500, 1500, True)):
        c = make_proj_case(rng, nq, nt)
        g = make_grid(0.0, 0.0, c["w"], c["h"])
        pred = np.clip(c["lvl"][:, 0] + 1, 0, 7).astype(np.int8)
        q_ur = t_ur = None
        if stereo:
            t_ur = np.where(rng.uniform(size=nt) < 0.6, c["tk"]["x"] - rng.uniform(5, 40, nt), -1).astype(np.float32)
            q_ur = (c["uvr"][:, 0] - rng.uniform(5, 40, nq)).astype(np.float32)
        gi, gd = m.window_best(c["uvr"], pred, c["qd"], c["qv"], g, c["tk"], c["td"], inv, chi2, q_ur, t_ur)
        wi, wd = oracle.window_best(c["uvr"], pred, c["qd"], c["qv"], c["gp"], c["tk"], c["start"], c["idx"], c["td"], inv, chi2, q_ur, t_ur)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
        assert (wi >= 0).sum() > 10


@pytest.mark.parametrize("ratio,ori,win", [(0.9, True, 100.0), (0.9, False, 100.0), (0.6, True, 30.0)])
def test_search_for_initialization_parity(gpu, oracle, ratio, ori, win):
    from matcher_cases import make_init_case
    from orbslamm_amd import ORBmatcher, make_grid
    rng = np.random.default_rng(int(ratio * 100 + win))
    total = 0
    for n1, n2 in ((500, 600), (2000, 2000), (50, 1200)):
        ic = make_init_case(rng, n1, n2)
        g = make_grid(0.0, 0.0, ic["w"], ic["h"])
        gm_, gn = ORBmatcher(ratio, ori, device=0).SearchForInitialization(ic["q_xy"], win, ic["k1"], ic["d1"], g, ic["k2"], ic["d2"])
        wm, wn = oracle.search_for_initialization(ic["q_xy"], win, ic["k1"], ic["d1"], ic["gp"], ic["k2"], ic["start"], ic["idx"],
                                                  ic["d2"], ratio, ori)
        assert gn == wn and np.array_equal(gm_, wm)
        total += wn
    assert total > 20


@pytest.mark.parametrize("ori", [True, False])
def test_search_for_triangulation_parity(gpu, oracle, ori):
    from matcher_cases import make_tri_case
    from orbslamm_amd import ORBmatcher
    rng = np.random.default_rng(300 + ori)
    total = 0
    for n1, n2, nn in ((400, 450, 19), (2000, 2000, 60), (300, 300, 1)):
        tc = make_tri_case(rng, n1, n2, nn)
        c = tc["c"]
        # rows of true partners made consistent so that some pairs pass the epipolar gate
        gm_, gn = ORBmatcher(0.6, ori, device=0).SearchForTriangulation(tc["k1"], c["qd"], 1 - c["qv"], c["qfv"], tc["k2"], c["td"],
                                                                        1 - c["tv"], c["tfv"], tc["F"], tc["ex"], tc["ey"], tc["sf2"], tc["sigma2"])
        wm, wn = oracle.search_for_triangulation(tc["k1"], c["qd"], 1 - c["qv"], c["qfv"], tc["k2"], c["td"], 1 - c["tv"], c["tfv"],
                                                 tc["F"], tc["ex"], tc["ey"], tc["sf2"], tc["sigma2"], False, ori)
        assert gn == wn and np.array_equal(gm_, wm)
        total += wn
    assert total > 100


def test_distinctive_descriptors_parity(gm, oracle):
    rng = np.random.default_rng(401)
    sizes = [0, 1, 2, 3, 7, 64, 65, 130] + rng.integers(1, 40, 300).tolist()
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = random_descriptors(rng, len(sizes))
    desc = np.concatenate([noisy_copies(rng, np.repeat(base[i:i + 1], s, axis=0), 20) if s else np.zeros((0, 32), np.uint8)
                           for i, s in enumerate(sizes)])
    got = gm.ComputeDistinctiveDescriptors(desc, start)
    want = oracle.distinctive_descriptors(desc, start)
    assert np.array_equal(got, want)
    assert got[0] == -1 and got[1] == 0


def test_cpp_adapters_on_gpu(gpu, tmp_path):
    """the C++ ORBextractor / ORBmatcher adapters (include/*.hpp) driven from a C++ program,
    checked bit-for-bit against the C oracle linked into the test binary"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oracle import binding as ob
    ob.build()
    exe = str(tmp_path / "adapter_gpu")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "adapter_gpu.cpp"), "-o", exe,
                           "-L", os.path.join(root, "orbslamm_amd"), "-lorbslamm_hip", "-L", os.path.join(root, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(root, "orbslamm_amd"), "-Wl,-rpath," + os.path.join(root, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter_gpu ok" in out.stdout


def test_extractor_dropin_reference_signature(gpu, tmp_path):
    """include/ORBextractor_hip.hpp under -DORBSLAMM_WITH_OPENCV: the reference's own operator() signature
    (InputArray, InputArray, vector<KeyPoint>&, OutputArray; include/ORBextractor.h:57-58) compiled against the
    data-holder cv:: types of tests/cpp/mock_opencv and called the way Frame::ExtractORB calls it (Frame.cc:247-253),
    keypoint records and descriptor rows against the C oracle; empty and featureless images (:1046-1047, :1064-1065)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oracle import binding as ob
    ob.build()
    exe = str(tmp_path / "adapter_cv_gpu")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-DORBSLAMM_WITH_OPENCV", "-I", os.path.join(root, "include"),
                           "-I", os.path.join(root, "tests", "cpp", "mock_opencv"),
                           os.path.join(root, "tests", "cpp", "adapter_cv_gpu.cpp"), "-o", exe,
                           "-L", os.path.join(root, "orbslamm_amd"), "-lorbslamm_hip", "-L", os.path.join(root, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(root, "orbslamm_amd"), "-Wl,-rpath," + os.path.join(root, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "adapter_cv_gpu ok" in out.stdout


def test_orbmatcher_dropin_all_eleven_signatures(gpu, tmp_path):
    """include/ORBmatcher_hip.hpp: ORBmatcherT<Frame, KeyFrame, MapPoint> -- the reference's eleven ORBmatcher members
    (ORBmatcher.h:48-83) instantiated on mock objects with the reference's member names, run on the GPU; the flattened
    arrays each member produced go through the C oracle (identical device results), the write-back into the object
    graph is replayed on a copy of the world, the projections are recomputed in double (tests/cpp/matcher_dropin_gpu.cpp)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oracle import binding as ob
    ob.build()
    exe = str(tmp_path / "matcher_dropin_gpu")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "matcher_dropin_gpu.cpp"), "-o", exe,
                           "-L", os.path.join(root, "orbslamm_amd"), "-lorbslamm_hip", "-L", os.path.join(root, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(root, "orbslamm_amd"), "-Wl,-rpath," + os.path.join(root, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "matcher_dropin_gpu ok" in out.stdout
    assert out.stdout.count("\n") >= 17  # one line per member variant


def test_undistort_keypoints_parity(gm, oracle):
    """TUM1.yaml calibration (fx 517.3, fy 516.5, cx 318.6, cy 255.3, k1 0.2624 k2 -0.9531 p1 -0.0054 p2 0.0026 k3 1.1633)"""
    rng = np.random.default_rng(501)
    n = 5000
    keys = np.zeros(n, dtype=oracle.KP_DTYPE)
    keys["x"] = rng.uniform(0, 640, n).astype(np.float32)
    keys["y"] = rng.uniform(0, 480, n).astype(np.float32)
    keys["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    keys["octave"] = rng.integers(0, 8, n)
    K = [517.306408, 516.469215, 318.643040, 255.313989]
    D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    got = gm.UndistortKeyPoints(keys, K, D)
    want = oracle.undistort_keypoints(keys, K, D)
    assert got.tobytes() == want.tobytes()
    assert np.abs(want["x"] - keys["x"]).max() > 1.0 and (want["angle"] == keys["angle"]).all()
    # zero distortion (KITTI): mvKeysUn = mvKeys
    assert gm.UndistortKeyPoints(keys, K, [0, 0, 0, 0, 0]).tobytes() == keys.tobytes()


def test_device_frame_feeds_projection_search(gpu, oracle):
    """extractor output stays in HBM -- the matcher builds the Frame's undistorted keys + grid from the
    device pointers and the projection searches use it as train side; same results as the oracle fed with the
    downloaded arrays and as the host-array entry point"""
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth
    w, h, nf = 640, 480, 1000
    K = [517.306408, 516.469215, 318.643040, 255.313989]
    D = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    fr = synth.make_frames(w, h, 2, stream=3)
    gex = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2, device=0)
    gex.extract_batch_device(*gex.upload_frames(fr))
    gex.sync()
    dk, dd, _, cap = gex.device_results()
    keys, desc = gex.download(1)
    nt = len(keys)
    rng = np.random.default_rng(8)
    for dist in (D, [0, 0, 0, 0, 0]):
        want_un = oracle.undistort_keypoints(keys, K, dist)
        # image bounds of the undistorted frame (Frame::ComputeImageBounds is the caller's; any box works for the test)
        x0, x1 = float(np.floor(want_un["x"].min())) - 1, float(np.ceil(want_un["x"].max())) + 1
        y0, y1 = float(np.floor(want_un["y"].min())) - 1, float(np.ceil(want_un["y"].max())) + 1
        g = make_grid(x0, y0, x1, y1)
        gp = oracle.make_grid_params(x0, y0, x1, y1)
        for ratio, ori, mode, th in ((0.9, True, 4, 100), (0.8, False, 3, 100), (0.75, True, 6, 50)):
            m = ORBmatcher(ratio, ori, device=0)
            frame = m.frame_from_device(dk + 1 * cap * 28, dd + 1 * cap * 32, nt, K, dist, g)
            got_un = m.frame_keys_un(frame)
            assert got_un.tobytes() == want_un.tobytes()
            start, idx = oracle.grid_build(gp, want_un)
            nq = 700
            src = rng.integers(0, nt, nq)
            qd = desc[src].copy()
            flip = rng.integers(0, 256, size=(nq, 10))
            for i in range(nq):
                for b in flip[i]:
                    qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
            uvr = np.zeros((nq, 3), np.float32)
            uvr[:, 0] = want_un["x"][src] + rng.normal(0, 3, nq)
            uvr[:, 1] = want_un["y"][src] + rng.normal(0, 3, nq)
            uvr[:, 2] = (15.0 * np.float32(1.2) ** want_un["octave"][src]).astype(np.float32)
            lvl = np.stack([want_un["octave"][src] - 1, want_un["octave"][src] + 1], axis=1).astype(np.int8)
            qa = (want_un["angle"][src] + rng.normal(0, 5, nq)).astype(np.float32) % np.float32(360)
            qv = (rng.uniform(size=nq) < 0.9).astype(np.uint8)
            qo = (rng.uniform(size=nq) < 0.7).astype(np.uint8)
            occ = (rng.uniform(size=nt) < 0.1).astype(np.uint8)
            a0 = np.full(nt, -1, np.int32)
            ga, gocc, gn = m.SearchByProjectionFrame(mode, th, uvr, lvl, qd, qa, qv, qo, frame, occ, a0)
            wa, wocc, wn = oracle.search_by_projection(mode, ratio, ori, th, uvr, lvl, qd, qa, qv, qo, gp, want_un, start, idx,
                                                       desc, occ, a0)
            ha, hocc, hn = m.SearchByProjection(mode, th, uvr, lvl, qd, qa, qv, qo, g, want_un, desc, occ, a0)
            assert gn == wn == hn and wn > 100
            assert np.array_equal(ga, wa) and np.array_equal(gocc, wocc) and np.array_equal(ga, ha)
            m.frame_destroy(frame)


def test_frame_settle_then_the_inputs_may_be_recycled(gpu, oracle):
    """orbm_frame_create returns with the build enqueued; orbm_frame_settle is what a caller that overwrites d_keys / d_desc
    at once waits on.  Here: a frame from two device buffers, settle, both buffers overwritten, then the search."""
    import ctypes as C
    from orbslamm_amd import ORBextractor, ORBmatcher, make_grid
    from orbslamm_amd._lib import check, ptr
    rng = np.random.default_rng(77)
    nq, nt = 900, 2500
    c = make_proj_case(rng, nq, nt)
    g = make_grid(0.0, 0.0, c["w"], c["h"])
    gex = ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=1, device=0)   # (its handle owns the raw device buffers)
    kb = np.frombuffer(c["tk"].tobytes(), np.uint8).reshape(1, 1, -1)
    db = np.ascontiguousarray(c["td"]).reshape(1, 1, -1)
    dk = gex.upload_frames(kb)[0]
    dd = gex.upload_frames(db)[0]
    m = ORBmatcher(0.9, True, device=0)
    frame = m.frame_from_device(dk, dd, nt, [500.0, 500.0, 320.0, 240.0], [0, 0, 0, 0, 0], g)
    m.frame_settle(frame)
    m.frame_settle(frame)   # (a second call returns at once)
    junk = np.full(max(kb.size, db.size), 0xA5, np.uint8)
    check(gex._L.orbx_upload(gex._h, C.c_void_p(dk), ptr(junk), C.c_size_t(kb.size)))
    check(gex._L.orbx_upload(gex._h, C.c_void_p(dd), ptr(junk), C.c_size_t(db.size)))
    a0 = np.full(nt, -1, np.int32)
    ga, gocc, gn = m.SearchByProjectionFrame(4, 100, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], frame, c["occ"], a0)
    wa, wocc, wn = oracle.search_by_projection(4, 0.9, True, 100, c["uvr"], c["lvl"], c["qd"], c["qa"], c["qv"], c["qo"], c["gp"],
                                               c["tk"], c["start"], c["idx"], c["td"], c["occ"], a0)
    assert gn == wn and np.array_equal(ga, wa) and np.array_equal(gocc, wocc) and wn > 100
    m.frame_destroy(frame)
    m.close(); gex.close()
