"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE -- see orb_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 16

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CORNER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("score", "<i4")])


class OrcExtractor(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_double), ("nlevels", C.c_int),
                ("iniThFAST", C.c_int), ("minThFAST", C.c_int),
                ("mvScaleFactor", C.c_float * MAX_LEVELS), ("mvInvScaleFactor", C.c_float * MAX_LEVELS),
                ("mvLevelSigma2", C.c_float * MAX_LEVELS), ("mvInvLevelSigma2", C.c_float * MAX_LEVELS),
                ("mnFeaturesPerLevel", C.c_int * MAX_LEVELS), ("umax", C.c_int * 16)]


class OrcGridParams(C.Structure):
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("invW", C.c_float), ("invH", C.c_float),
                ("cols", C.c_int), ("rows", C.c_int)]


class OrcFeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("node_id", C.c_void_p), ("start", C.c_void_p), ("idx", C.c_void_p)]


class OrcProjParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("nnratio", C.c_float), ("check_ori", C.c_int), ("th_dist", C.c_int)]


def build(march_native=False, out_dir=None):
    """compile the oracle; returns the path of the .so"""
    out_dir = out_dir or _HERE
    so = os.path.join(out_dir, "liborb_oracle_native.so" if march_native else "liborb_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orb_extract.c", "orb_match.c", "orb_vocab.c", "orb_stereo.c")]
    deps = srcs + [os.path.join(_HERE, f) for f in ("orb_oracle.h", "brief_pattern.inc")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    cmd = ["gcc", "-O3", "-fPIC", "-std=c11", "-ffp-contract=off", "-fno-fast-math"]
    if march_native:
        cmd.append("-march=native")
    cmd += ["-shared", "-o", so] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return so


_lib = None


def lib(path=None):
    global _lib
    if _lib is None or path is not None:
        so = path or os.path.join(_HERE, "liborb_oracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.orc_extractor_init.restype = C.c_int
        L.orc_extract.restype = C.c_int
        L.orc_fast9_16.restype = C.c_int
        L.orc_level_candidates.restype = C.c_int
        L.orc_distribute.restype = C.c_int
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_rot_bin.restype = C.c_int
        L.orc_rot_bin.argtypes = [C.c_float, C.c_float]
        L.orc_match_bruteforce.restype = C.c_int
        L.orc_features_in_area.restype = C.c_int
        L.orc_search_by_bow.restype = C.c_int
        L.orc_search_by_projection.restype = C.c_int
        L.orc_window_best.restype = C.c_int
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_triangulation.restype = C.c_int
        L.orc_vocab_create.restype = C.c_void_p
        L.orc_vocab_transform.restype = C.c_int
        if path is not None:
            return L
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Extractor:
    """mirror of ORBextractor (reference include/ORBextractor.h:45-111) on the oracle"""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, L=None):
        self.L = L or lib()
        self.ex = OrcExtractor()
        rc = self.L.orc_extractor_init(C.byref(self.ex), int(nfeatures), C.c_float(scaleFactor),
                                       int(nlevels), int(iniThFAST), int(minThFAST))
        if rc != 0:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def scale_factors(self):
        return np.array(self.ex.mvScaleFactor[:self.nlevels], dtype=np.float32)

    def features_per_level(self):
        return list(self.ex.mnFeaturesPerLevel[:self.nlevels])

    def umax(self):
        return list(self.ex.umax)

    def level_size(self, w, h, level):
        lw, lh = C.c_int(), C.c_int()
        self.L.orc_level_size(C.byref(self.ex), w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    def __call__(self, img, want_pyramid=False):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 8 * self.nlevels + 64
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        cand = np.zeros(MAX_LEVELS, dtype=np.int32)
        kept = np.zeros(MAX_LEVELS, dtype=np.int32)
        pyr = None
        if want_pyramid:
            tot = sum(a * b for a, b in (self.level_size(w, h, l) for l in range(self.nlevels)))
            pyr = np.zeros(tot, dtype=np.uint8)
        n = self.L.orc_extract(C.byref(self.ex), _p(img), w, h, w, _p(kps), _p(desc), cap,
                               _p(pyr), _p(cand), _p(kept))
        if n < 0:
            raise RuntimeError("orc_extract failed: %d" % n)
        out = dict(kps=kps[:n].copy(), desc=desc[:n].copy(), cand_counts=cand[:self.nlevels].copy(),
                   kept_counts=kept[:self.nlevels].copy())
        if want_pyramid:
            out["pyramid"] = pyr
        return out

    def cell_stats(self, img):
        """(cells, cells run again at minThFAST, cells empty at both thresholds) over all levels of one frame (ref:808-816)"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        pyr = self(img, want_pyramid=True)["pyramid"]
        tot, off = [0, 0, 0], 0
        for l in range(self.nlevels):
            lw, lh = self.level_size(w, h, l)
            lvl = np.ascontiguousarray(pyr[off:off + lw * lh].reshape(lh, lw))
            off += lw * lh
            a, b, c = C.c_int(), C.c_int(), C.c_int()
            self.L.orc_level_cell_stats(C.byref(self.ex), _p(lvl), lw, lh, lw, C.byref(a), C.byref(b), C.byref(c))
            tot[0] += a.value; tot[1] += b.value; tot[2] += c.value
        return tuple(tot)

    def level_candidates(self, level_img):
        level_img = np.ascontiguousarray(level_img, dtype=np.uint8)
        h, w = level_img.shape
        cap = w * h // 4 + 16
        out = np.zeros(cap, dtype=CORNER_DTYPE)
        n = self.L.orc_level_candidates(C.byref(self.ex), _p(level_img), w, h, w, _p(out), cap)
        if n < 0:
            raise RuntimeError("candidate overflow")
        return out[:n].copy()


def resize(src, dw, dh, L=None):
    L = L or lib()
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    L.orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def fast(img, threshold, L=None):
    L = L or lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros(w * h // 4 + 16, dtype=CORNER_DTYPE)
    n = L.orc_fast9_16(_p(img), w, h, w, int(threshold), _p(out), out.shape[0])
    return out[:n].copy()


def gaussian7(img, L=None):
    L = L or lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dst = np.zeros_like(img)
    L.orc_gaussian7_u8(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(dst), img.shape[1])
    return dst


def distribute(cands, minX, maxX, minY, maxY, N, L=None):
    L = L or lib()
    cands = np.ascontiguousarray(cands, dtype=CORNER_DTYPE)
    out = np.zeros(N + 64, dtype=CORNER_DTYPE)
    n = L.orc_distribute(_p(cands), cands.shape[0], minX, maxX, minY, maxY, N, _p(out), out.shape[0])
    if n < 0:
        raise RuntimeError("orc_distribute failed %d" % n)
    return out[:n].copy()


def descriptor_distance(a, b, L=None):
    L = L or lib()
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return L.orc_descriptor_distance(_p(a), _p(b))


def match_bruteforce(qdesc, qangle, tdesc, tangle, nnratio=0.7, th_low=50, check_ori=True, L=None):
    L = L or lib()
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    qangle = np.ascontiguousarray(qangle, dtype=np.float32)
    tangle = np.ascontiguousarray(tangle, dtype=np.float32)
    nq, nt = qdesc.shape[0], tdesc.shape[0]
    match = np.full(max(nq, 1), -1, dtype=np.int32)
    n = L.orc_match_bruteforce(_p(qdesc), _p(qangle), nq, _p(tdesc), _p(tangle), nt,
                               C.c_float(nnratio), int(th_low), int(bool(check_ori)), _p(match))
    return match[:nq], n


def make_grid_params(minX, minY, maxX, maxY, cols=64, rows=48):
    gp = OrcGridParams()
    gp.minX, gp.minY = minX, minY
    gp.invW = np.float32(cols) / np.float32(np.float32(maxX) - np.float32(minX))
    gp.invH = np.float32(rows) / np.float32(np.float32(maxY) - np.float32(minY))
    gp.cols, gp.rows = cols, rows
    return gp


def grid_build(gp, keys_un, L=None):
    L = L or lib()
    keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
    n = keys_un.shape[0]
    start = np.zeros(gp.cols * gp.rows + 1, dtype=np.int32)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    L.orc_grid_build(C.byref(gp), _p(keys_un), n, _p(start), _p(idx))
    return start, idx


def features_in_area(gp, keys_un, start, idx, x, y, r, minLevel=-1, maxLevel=-1, L=None):
    L = L or lib()
    keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
    out = np.zeros(max(keys_un.shape[0], 1), dtype=np.int32)
    n = L.orc_features_in_area(C.byref(gp), _p(keys_un), _p(start), _p(idx), C.c_float(x), C.c_float(y),
                               C.c_float(r), int(minLevel), int(maxLevel), _p(out), out.shape[0])
    return out[:n].copy()


def _featvec(node_id, start, idx):
    fv = OrcFeatVec()
    keep = (np.ascontiguousarray(node_id, dtype=np.uint32), np.ascontiguousarray(start, dtype=np.int32),
            np.ascontiguousarray(idx, dtype=np.int32))
    fv.n_nodes = keep[0].shape[0]
    fv.node_id, fv.start, fv.idx = (k.ctypes.data for k in keep)
    return fv, keep


def search_by_bow(qdesc, qangle, qvalid, qfv, tdesc, tangle, tvalid, tfv, nnratio, check_ori,
                  out_by_train, L=None):
    """qfv / tfv: (node_id, start, idx) CSR triples"""
    L = L or lib()
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    qangle = np.ascontiguousarray(qangle, dtype=np.float32)
    tangle = np.ascontiguousarray(tangle, dtype=np.float32)
    qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
    tv = None if tvalid is None else np.ascontiguousarray(tvalid, dtype=np.uint8)
    nq, nt = qdesc.shape[0], tdesc.shape[0]
    fq, kq = _featvec(*qfv)
    ft, kt = _featvec(*tfv)
    match = np.full(max(nt if out_by_train else nq, 1), -1, dtype=np.int32)
    n = L.orc_search_by_bow(_p(qdesc), _p(qangle), _p(qv), nq, C.byref(fq),
                            _p(tdesc), _p(tangle), _p(tv), nt, C.byref(ft),
                            C.c_float(nnratio), int(bool(check_ori)), int(bool(out_by_train)), _p(match))
    return match[:(nt if out_by_train else nq)], n


def search_by_projection(mode, nnratio, check_ori, th_dist, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos,
                         gp, t_keys_un, start, idx, tdesc, t_occ, assign, L=None):
    L = L or lib()
    pp = OrcProjParams(int(mode), float(nnratio), int(bool(check_ori)), int(th_dist))
    q_uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
    q_lvl = np.ascontiguousarray(q_lvl, dtype=np.int8)
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    qangle = np.ascontiguousarray(qangle, dtype=np.float32)
    qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
    qo = None if q_obs_pos is None else np.ascontiguousarray(q_obs_pos, dtype=np.uint8)
    t_keys_un = np.ascontiguousarray(t_keys_un, dtype=KP_DTYPE)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    t_occ = np.ascontiguousarray(t_occ, dtype=np.uint8).copy()
    assign = np.ascontiguousarray(assign, dtype=np.int32).copy()
    n = L.orc_search_by_projection(C.byref(pp), _p(q_uvr), _p(q_lvl), _p(qdesc), _p(qangle), _p(qv), _p(qo),
                                   q_uvr.shape[0], C.byref(gp), _p(t_keys_un), _p(start), _p(idx),
                                   _p(tdesc), t_keys_un.shape[0], _p(t_occ), _p(assign))
    return assign, t_occ, n


def window_best(q_uvr, q_pred, qdesc, qvalid, gp, t_keys_un, start, idx, tdesc, inv_sigma2=None, chi2=False,
                q_ur=None, t_uright=None, L=None):
    L = L or lib()
    q_uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
    q_pred = np.ascontiguousarray(q_pred, dtype=np.int8)
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
    t_keys_un = np.ascontiguousarray(t_keys_un, dtype=KP_DTYPE)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    sig = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, dtype=np.float32)
    qur = None if q_ur is None else np.ascontiguousarray(q_ur, dtype=np.float32)
    tur = None if t_uright is None else np.ascontiguousarray(t_uright, dtype=np.float32)
    nq = q_uvr.shape[0]
    bi = np.full(max(nq, 1), -1, np.int32)
    bd = np.full(max(nq, 1), 256, np.int32)
    L.orc_window_best(_p(q_uvr), _p(qur), _p(q_pred), _p(qdesc), _p(qv), nq, C.byref(gp), _p(t_keys_un), _p(start), _p(idx),
                      _p(tdesc), _p(tur), t_keys_un.shape[0], _p(sig), int(bool(chi2)), _p(bi), _p(bd))
    return bi[:nq], bd[:nq]


def search_for_initialization(q_xy, window, q_keys_un, qdesc, gp, t_keys_un, start, idx, tdesc, nnratio, check_ori, L=None):
    L = L or lib()
    q_xy = np.ascontiguousarray(q_xy, dtype=np.float32)
    q_keys_un = np.ascontiguousarray(q_keys_un, dtype=KP_DTYPE)
    qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
    t_keys_un = np.ascontiguousarray(t_keys_un, dtype=KP_DTYPE)
    tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
    nq = q_keys_un.shape[0]
    m12 = np.full(max(nq, 1), -1, np.int32)
    n = L.orc_search_for_initialization(_p(q_xy), C.c_float(window), _p(q_keys_un), _p(qdesc), nq, C.byref(gp), _p(t_keys_un),
                                        _p(start), _p(idx), _p(tdesc), t_keys_un.shape[0], C.c_float(nnratio),
                                        int(bool(check_ori)), _p(m12))
    return m12[:nq], n


def search_for_triangulation(k1, d1, skip1, fv1, k2, d2, skip2, fv2, F12, ex, ey, sf2, sigma2_2, only_stereo, check_ori,
                             uright1=None, uright2=None, L=None):
    L = L or lib()
    k1 = np.ascontiguousarray(k1, dtype=KP_DTYPE)
    k2 = np.ascontiguousarray(k2, dtype=KP_DTYPE)
    d1 = np.ascontiguousarray(d1, dtype=np.uint8)
    d2 = np.ascontiguousarray(d2, dtype=np.uint8)
    s1 = None if skip1 is None else np.ascontiguousarray(skip1, dtype=np.uint8)
    s2 = None if skip2 is None else np.ascontiguousarray(skip2, dtype=np.uint8)
    u1 = None if uright1 is None else np.ascontiguousarray(uright1, dtype=np.float32)
    u2 = None if uright2 is None else np.ascontiguousarray(uright2, dtype=np.float32)
    F = np.ascontiguousarray(F12, dtype=np.float32).reshape(9)
    sf2 = np.ascontiguousarray(sf2, dtype=np.float32)
    sg2 = np.ascontiguousarray(sigma2_2, dtype=np.float32)
    f1, keep1 = _featvec(*fv1)
    f2, keep2 = _featvec(*fv2)
    n1 = k1.shape[0]
    m12 = np.full(max(n1, 1), -1, np.int32)
    n = L.orc_search_for_triangulation(_p(k1), _p(d1), _p(s1), _p(u1), n1, C.byref(f1), _p(k2), _p(d2), _p(s2), _p(u2),
                                       k2.shape[0], C.byref(f2), _p(F), C.c_float(ex), C.c_float(ey), _p(sf2), _p(sg2),
                                       int(bool(only_stereo)), int(bool(check_ori)), _p(m12))
    return m12[:n1], n


class Vocabulary:
    """DBoW2 vocabulary on the oracle (nodes in loadFromTextFile order)"""

    def __init__(self, k, Lv, scoring, weighting, parent, is_leaf, desc, weight, L=None):
        self.L = L or lib()
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        self.v = C.c_void_p(self.L.orc_vocab_create(int(k), int(Lv), int(scoring), int(weighting), parent.shape[0],
                                                    _p(parent), _p(is_leaf), _p(desc), _p(weight)))
        if not self.v.value:
            raise ValueError("bad vocabulary")

    def __del__(self):
        try:
            self.L.orc_vocab_free(self.v)
        except Exception:
            pass

    def transform(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        n = desc.shape[0]
        wid = np.zeros(max(n, 1), np.uint32)
        ww = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32)
        fs = np.zeros(n + 1, np.int32)
        fi = np.zeros(max(n, 1), np.int32)
        nw, nf = C.c_int(), C.c_int()
        self.L.orc_vocab_transform(self.v, _p(desc), n, int(levelsup), _p(wid), _p(ww), C.byref(nw), _p(fn), _p(fs), _p(fi), C.byref(nf))
        return (wid[:nw.value].copy(), ww[:nw.value].copy()), (fn[:nf.value].copy(), fs[:nf.value + 1].copy(), fi[:fs[nf.value]].copy())


    def transform_one(self, desc, levelsup):
        """per feature: (word id, weight, node id at L - levelsup) -- the stream transform() aggregates"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        n = desc.shape[0]
        word = np.zeros(n, np.uint32); w = np.zeros(n, np.float64); nid = np.zeros(n, np.uint32)
        a, b, c = C.c_uint32(), C.c_double(), C.c_uint32()
        self.L.orc_vocab_transform_one.restype = None
        for i in range(n):
            self.L.orc_vocab_transform_one(self.v, desc[i].ctypes.data_as(C.c_void_p), int(levelsup), C.byref(a), C.byref(b), C.byref(c))
            word[i], w[i], nid[i] = a.value, b.value, c.value
        return word, w, nid


# ---------------------------------------------------------------- oracle/_ref: the reference's own object code
REF_SO = os.path.join(_HERE, "_ref", "libdbow2_ref.so")


class OrcPyramid(C.Structure):
    _fields_ = [("nlevels", C.c_int), ("data", C.c_void_p * 16), ("w", C.c_int * 16), ("h", C.c_int * 16), ("stride", C.c_int * 16)]


def compute_stereo_matches(keysL, descL, keysR, descR, levelsL, levelsR, scale, inv_scale, mb, mbf, L=None):
    """Frame::ComputeStereoMatches (Frame.cc:466-638); levelsL / levelsR: lists of contiguous uint8 images (the
    un-blurred pyramid levels).  Returns (mvuRight, mvDepth, accepted before the median filter)."""
    L = L or lib()
    keysL = np.ascontiguousarray(keysL); keysR = np.ascontiguousarray(keysR)
    descL = np.ascontiguousarray(descL, dtype=np.uint8); descR = np.ascontiguousarray(descR, dtype=np.uint8)

    def pyr(levels):
        p = OrcPyramid()
        p.nlevels = len(levels)
        keep = []
        for i, im in enumerate(levels):
            im = np.ascontiguousarray(im, dtype=np.uint8)
            keep.append(im)
            p.data[i] = im.ctypes.data
            p.h[i], p.w[i] = im.shape
            p.stride[i] = im.shape[1]
        return p, keep
    pl, k1 = pyr(levelsL)
    pr, k2 = pyr(levelsR)
    sc = np.ascontiguousarray(scale, dtype=np.float32); isc = np.ascontiguousarray(inv_scale, dtype=np.float32)
    ur = np.zeros(max(len(keysL), 1), dtype=np.float32); dp = np.zeros(max(len(keysL), 1), dtype=np.float32)
    L.orc_compute_stereo_matches.restype = C.c_int
    n = L.orc_compute_stereo_matches(_p(keysL), _p(descL), len(keysL), _p(keysR), _p(descR), len(keysR), C.byref(pl), C.byref(pr),
                                     _p(sc), _p(isc), C.c_float(mb), C.c_float(mbf), _p(ur), _p(dp))
    return ur[:len(keysL)], dp[:len(keysL)], n


def build_ref():
    """oracle/Makefile target _ref: DBoW2 BowVector.cpp + FeatureVector.cpp compiled from /root/reference where they
    lie (no-op without the checkout: the GPU box only has the prebuilt library).  Returns the path or None."""
    subprocess.call(["make", "-s", "-C", _HERE, "_ref"])
    return REF_SO if os.path.exists(REF_SO) else None


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        so = REF_SO if os.path.exists(REF_SO) else build_ref()
        if so is None:
            return None
        _ref = C.CDLL(so)
    return _ref


def ref_bow_build(ids, w, add_if_not_exist, norm):
    """DBoW2::BowVector fed (ids[i], w[i]) in order; norm: 0 none, 1 L1, 2 L2 -> (ids, values) in map order"""
    R = ref_lib()
    ids = np.ascontiguousarray(ids, dtype=np.uint32); w = np.ascontiguousarray(w, dtype=np.float64)
    oid = np.zeros(max(len(ids), 1), np.uint32); ow = np.zeros(max(len(ids), 1), np.float64)
    m = R.ref_bow_build(_p(ids), _p(w), len(ids), int(bool(add_if_not_exist)), int(norm), _p(oid), _p(ow))
    return oid[:m].copy(), ow[:m].copy()


def ref_fv_build(node, feat):
    """DBoW2::FeatureVector fed addFeature(node[i], feat[i]) in order -> CSR (nodes, start, idx) in map order"""
    R = ref_lib()
    node = np.ascontiguousarray(node, dtype=np.uint32); feat = np.ascontiguousarray(feat, dtype=np.uint32)
    on = np.zeros(max(len(node), 1), np.uint32); st = np.zeros(len(node) + 1, np.int32); ix = np.zeros(max(len(node), 1), np.int32)
    k = R.ref_fv_build(_p(node), _p(feat), len(node), _p(on), _p(st), _p(ix))
    return on[:k].copy(), st[:k + 1].copy(), ix[:st[k]].copy()


def distinctive_descriptors(desc, start, L=None):
    L = L or lib()
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    start = np.ascontiguousarray(start, dtype=np.int32)
    out = np.full(max(len(start) - 1, 1), -1, np.int32)
    L.orc_distinctive_descriptors(_p(desc), _p(start), len(start) - 1, _p(out))
    return out[:len(start) - 1]


def undistort_keypoints(keys, K, D, L=None):
    L = L or lib()
    keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
    K = np.ascontiguousarray(K, dtype=np.float32)
    D = np.ascontiguousarray(D, dtype=np.float32)
    out = np.zeros_like(keys)
    L.orc_undistort_keypoints(_p(keys), keys.shape[0], _p(K), _p(D), _p(out))
    return out


def stereo_from_rgbd(keys, keys_un, depth, mbf, L=None):
    """Frame::ComputeStereoFromRGBD (Frame.cc:641-663) -> (mvuRight, mvDepth)"""
    L = L or lib()
    keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
    keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    n = keys.shape[0]
    ur = np.zeros(n, np.float32); dp = np.zeros(n, np.float32)
    L.orc_stereo_from_rgbd(_p(keys), _p(keys_un), n, _p(depth), depth.shape[1], depth.shape[0], depth.shape[1], C.c_float(mbf), _p(ur), _p(dp))
    return ur, dp
