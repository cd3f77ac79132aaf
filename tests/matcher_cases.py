"""Shared synthetic case builders for the matcher tests (CPU and GPU)."""
import numpy as np


def noisy_copies(rng, base, nflip):
    q = base.copy()
    for i in range(len(q)):
        for b in rng.integers(0, 256, nflip):
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q


def naive_bruteforce(q, qa, t, ta, nnratio, th_low, check_ori):
    """independent numpy re-derivation of the headline rule"""
    nq, nt = len(q), len(t)
    match = np.full(nq, -1, np.int32)
    if nt == 0:
        return match, 0
    tb = np.unpackbits(t, axis=1)
    bins = {}
    for i in range(nq):
        d = (np.unpackbits(q[i])[None, :] != tb).sum(axis=1)
        order = np.argsort(d, kind="stable")
        b1 = int(d[order[0]])
        b2 = int(d[order[1]]) if nt > 1 else 256
        if b1 <= th_low and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
            match[i] = order[0]
            if check_ori:
                rot = np.float32(qa[i]) - np.float32(ta[order[0]])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                v = np.float32(rot * np.float32(1.0 / 30))
                b = int(np.floor(v + np.float32(0.5))) if v >= 0 else int(np.ceil(v - np.float32(0.5)))
                if b == 30:
                    b = 0
                bins.setdefault(b, []).append(i)
    if check_ori and bins:
        sizes = [len(bins.get(b, [])) for b in range(30)]
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i, s in enumerate(sizes):
            if s > m1:
                m3, m2, m1 = m2, m1, s
                i3, i2, i1 = i2, i1, i
            elif s > m2:
                m3, m2 = m2, s
                i3, i2 = i2, i
            elif s > m3:
                m3, i3 = s, i
        if np.float32(m2) < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif np.float32(m3) < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for b, lst in bins.items():
            if b not in (i1, i2, i3):
                for i in lst:
                    match[i] = -1
    return match, int((match >= 0).sum())


def make_bow_case(rng, nq, nt, nnodes):
    """KeyFrame (queries) vs Frame (trains) with synthetic FeatureVectors: features are
    spread over `nnodes` vocabulary nodes; some nodes exist on one side only."""
    n = max(nq, nt)
    base = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    qd = noisy_copies(rng, base[:nq], 10)
    perm = rng.permutation(nt)
    td = base[:nt][perm] if nt <= n else base
    node_of_base = rng.integers(0, nnodes, n)
    qnode = node_of_base[:nq].copy()
    tnode = node_of_base[:nt][perm].copy()
    # a few features land in side-only nodes
    qnode[rng.integers(0, nq, max(1, nq // 20))] = nnodes + 1
    tnode[rng.integers(0, nt, max(1, nt // 20))] = nnodes + 2

    def csr(node, count):
        ids = np.unique(node)
        start = [0]
        idx = []
        for i in ids:
            members = np.nonzero(node == i)[0]
            members = members[rng.permutation(len(members))]  # stored order is arbitrary
            idx.extend(members.tolist())
            start.append(len(idx))
        return (ids.astype(np.uint32) * 7 + 3, np.array(start, np.int32), np.array(idx, np.int32))

    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    inv = np.argsort(perm)
    for i in range(min(nq, nt)):
        ta[inv[i]] = np.float32((qa[i] + rng.normal(0, 4)) % 360)
    return dict(qd=qd, td=td, qa=qa, ta=ta, qv=(rng.uniform(size=nq) < 0.85).astype(np.uint8),
                tv=(rng.uniform(size=nt) < 0.9).astype(np.uint8), qfv=csr(qnode, nq), tfv=csr(tnode, nt))


def make_proj_case(rng, nq, nt, w=1241.0, h=376.0):
    """train frame with keypoints/grid, queries projected near some of them"""
    from oracle import binding as ob
    tk = np.zeros(nt, dtype=ob.KP_DTYPE)
    tk["x"] = rng.uniform(0, w, nt).astype(np.float32)
    tk["y"] = rng.uniform(0, h, nt).astype(np.float32)
    tk["octave"] = rng.integers(0, 8, nt)
    tk["angle"] = rng.uniform(0, 360, nt).astype(np.float32)
    td = rng.integers(0, 256, size=(nt, 32), dtype=np.uint8)
    src = rng.integers(0, nt, nq)
    qd = noisy_copies(rng, td[src], 14)
    uvr = np.zeros((nq, 3), np.float32)
    uvr[:, 0] = tk["x"][src] + rng.normal(0, 4, nq)
    uvr[:, 1] = tk["y"][src] + rng.normal(0, 4, nq)
    sf = np.float32(1.2) ** tk["octave"][src]
    uvr[:, 2] = (15.0 * sf).astype(np.float32)
    lvl = np.stack([tk["octave"][src] - 1, tk["octave"][src] + 1], axis=1).astype(np.int8)
    qa = (tk["angle"][src] + rng.normal(0, 5, nq)).astype(np.float32) % np.float32(360)
    gp = ob.make_grid_params(0.0, 0.0, w, h)
    start, idx = ob.grid_build(gp, tk)
    return dict(uvr=uvr, lvl=lvl, qd=qd, qa=qa, qv=(rng.uniform(size=nq) < 0.9).astype(np.uint8),
                qo=(rng.uniform(size=nq) < 0.7).astype(np.uint8), gp=gp, tk=tk, start=start, idx=idx, td=td,
                occ=(rng.uniform(size=nt) < 0.1).astype(np.uint8), w=w, h=h)


def make_init_case(rng, n1, n2, w=640.0, h=480.0):
    """two frames for SearchForInitialization: F2 = F1 shifted by a few pixels"""
    from oracle import binding as ob
    k1 = np.zeros(n1, dtype=ob.KP_DTYPE)
    k1["x"] = rng.uniform(20, w - 20, n1).astype(np.float32)
    k1["y"] = rng.uniform(20, h - 20, n1).astype(np.float32)
    k1["octave"] = (rng.uniform(size=n1) < 0.4) * rng.integers(1, 8, n1)  # 60 % on octave 0
    k1["angle"] = rng.uniform(0, 360, n1).astype(np.float32)
    d1 = rng.integers(0, 256, size=(n1, 32), dtype=np.uint8)
    src = rng.integers(0, n1, n2)
    k2 = np.zeros(n2, dtype=ob.KP_DTYPE)
    k2["x"] = np.clip(k1["x"][src] + rng.normal(0, 12, n2), 0, w - 1).astype(np.float32)
    k2["y"] = np.clip(k1["y"][src] + rng.normal(0, 12, n2), 0, h - 1).astype(np.float32)
    k2["octave"] = k1["octave"][src]
    k2["angle"] = ((k1["angle"][src] + rng.normal(0, 5, n2)) % 360).astype(np.float32)
    d2 = noisy_copies(rng, d1[src], 12)
    gp = ob.make_grid_params(0.0, 0.0, w, h)
    start, idx = ob.grid_build(gp, k2)
    q_xy = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    return dict(k1=k1, d1=d1, k2=k2, d2=d2, gp=gp, start=start, idx=idx, q_xy=q_xy, w=w, h=h)


def make_tri_case(rng, n1, n2, nnodes, w=640.0, h=480.0):
    """two keyframes related by a pure x-translation: F12 = [t]x, epipolar lines are image rows"""
    from oracle import binding as ob
    c = make_bow_case(rng, n1, n2, nnodes)
    k1 = np.zeros(n1, dtype=ob.KP_DTYPE)
    k2 = np.zeros(n2, dtype=ob.KP_DTYPE)
    k1["x"] = rng.uniform(0, w, n1).astype(np.float32)
    k1["y"] = rng.uniform(0, h, n1).astype(np.float32)
    k1["angle"] = c["qa"]
    k1["octave"] = rng.integers(0, 8, n1)
    k2["x"] = rng.uniform(0, w, n2).astype(np.float32)
    k2["y"] = rng.uniform(0, h, n2).astype(np.float32)
    k2["angle"] = c["ta"]
    k2["octave"] = rng.integers(0, 8, n2)
    # true correspondences (nearest descriptor) lie near the same row: noise partly inside,
    # partly outside the chi-square gate of the epipolar test
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint8)
    for i in range(n1):
        d = lut[np.bitwise_xor(c["qd"][i][None, :], c["td"])].sum(axis=1)
        j = int(np.argmin(d))
        if d[j] <= 50:
            k2["y"][j] = np.float32(k1["y"][i] + rng.normal(0, 2.5))
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # l = (0, -1, y1): |y2 - y1| distance
    return dict(c=c, k1=k1, k2=k2, F=F, ex=np.float32(-5000.0), ey=np.float32(h / 2), sf2=sf, sigma2=(sf * sf).astype(np.float32))
