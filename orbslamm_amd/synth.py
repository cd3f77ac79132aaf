"""Deterministic synthetic camera streams: a static scene of random
rectangles and discs on a canvas, a slowly panning viewport, additive noise.
Pure numpy so that every harness (CPU checker, GPU bench) sees identical bytes."""
import numpy as np


def _tri(a, m):
    a = a % (2 * m)
    return a if a <= m else 2 * m - a


def make_scene(w, h, stream=0, nshapes=None):
    cw, ch = w + 64, h + 16
    if nshapes is None:
        nshapes = max(200, int(4000 * (w * h) / (1241.0 * 376.0)))
    rng = np.random.Generator(np.random.PCG64(0x0B5A4000 + stream))
    canvas = np.full((ch, cw), 128, dtype=np.uint8)
    cx = rng.integers(0, cw, nshapes)
    cy = rng.integers(0, ch, nshapes)
    sz = rng.integers(4, 41, nshapes)
    sz2 = rng.integers(4, 41, nshapes)
    val = rng.integers(0, 256, nshapes)
    kind = rng.integers(0, 2, nshapes)
    for i in range(nshapes):
        x0, y0 = int(cx[i]), int(cy[i])
        if kind[i] == 0:
            canvas[max(0, y0 - sz2[i] // 2):y0 + sz2[i] // 2 + 1, max(0, x0 - sz[i] // 2):x0 + sz[i] // 2 + 1] = val[i]
        else:
            r = int(sz[i]) // 2
            ya, yb = max(0, y0 - r), min(ch, y0 + r + 1)
            xa, xb = max(0, x0 - r), min(cw, x0 + r + 1)
            yy, xx = np.ogrid[ya:yb, xa:xb]
            m = (yy - y0) ** 2 + (xx - x0) ** 2 <= r * r
            canvas[ya:yb, xa:xb][m] = val[i]
    return canvas


def frame_from_scene(canvas, w, h, t, stream=0):
    ox, oy = _tri(2 * t, 64), _tri(t, 16)
    view = canvas[oy:oy + h, ox:ox + w].astype(np.int16)
    rng = np.random.Generator(np.random.PCG64((0x5EED0000 + stream) * 100003 + t))
    noise = rng.integers(-4, 5, size=(h, w), dtype=np.int16)
    return np.clip(view + noise, 0, 255).astype(np.uint8)


def make_frames(w, h, nframes, stream=0, t0=0):
    canvas = make_scene(w, h, stream)
    return np.stack([frame_from_scene(canvas, w, h, t0 + t, stream) for t in range(nframes)])


def make_vocabulary(k, L, seed=7, ragged=False):
    """A k-ary vocabulary tree of depth L in DBoW2's loadFromTextFile order (breadth first; the real ORBvoc.txt -- k = 10,
    L = 6 -- is absent from the reference checkout): children are their parent's descriptor with a few bits flipped.
    ragged=False: complete (k^l nodes at level l).  ragged=True: strongly ragged -- below the root a node has 2 .. k children
    (k-means clusters that came out empty are not stored) and from level 2 on a tenth of the nodes are leaves early (clusters
    too small to split), so words sit at different depths.  ragged=p (a float in (0, 1)): every child exists with probability
    p and a node below level 1 is an early leaf with probability 1 - p: p = 0.994 gives ~1.08 M nodes at k = 10, L = 6, the
    size of the published ORBvoc.txt (1 082 073 nodes, 97 % of the complete tree's 1 111 110).
    Returns dict(parent, is_leaf, desc, weight, k, L) in the layout orbv_create takes."""
    rng = np.random.default_rng(seed)
    parent, desc, is_leaf = [], [], []
    prev_ids = np.array([0])
    prev_desc = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    next_id = 1
    for lvl in range(1, L + 1):
        if not ragged or lvl == 1:
            nch = np.full(len(prev_ids), k)
        elif ragged is True:
            nch = rng.integers(2, k + 1, len(prev_ids))
        else:
            nch = np.maximum(rng.binomial(k, float(ragged), len(prev_ids)), 1)
        n = int(nch.sum())
        d = np.repeat(prev_desc, nch, axis=0)
        nflip = max(4, 60 >> (lvl - 1))
        bits = rng.integers(0, 256, (n, nflip))
        for j in range(nflip):
            np.bitwise_xor.at(d, (np.arange(n), bits[:, j] >> 3), (1 << (bits[:, j] & 7)).astype(np.uint8))
        leaf = np.full(n, lvl == L)
        if ragged and 2 <= lvl < L:
            leaf |= rng.uniform(size=n) < (0.1 if ragged is True else 1.0 - float(ragged))
        parent.append(np.repeat(prev_ids, nch))
        desc.append(d)
        is_leaf.append(leaf.astype(np.uint8))
        ids = np.arange(next_id, next_id + n)
        prev_ids, prev_desc = ids[~leaf], d[~leaf]
        next_id += n
        if len(prev_ids) == 0:
            break
    w = rng.uniform(0.1, 9.0, next_id - 1)
    return dict(parent=np.concatenate(parent).astype(np.int32), is_leaf=np.concatenate(is_leaf), desc=np.concatenate(desc),
                weight=w, k=k, L=L)
