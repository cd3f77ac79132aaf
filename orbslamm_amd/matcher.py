"""Host-side mirror of the reference's ORBmatcher (include/ORBmatcher.h:37-102) over
the C ABI, on flattened arrays (the MapPoint/KeyFrame object graph stays with the
caller).  All Hamming work runs in HIP kernels."""
import ctypes as C

import numpy as np

from ._lib import KP_DTYPE, OrbmFeatVec, OrbmGrid, OrbmProjParams, check, lib, ptr


def make_grid(minX, minY, maxX, maxY, cols=64, rows=48):
    """Frame.cc:212-213: mfGridElementWidthInv = FRAME_GRID_COLS / (mnMaxX - mnMinX) in float"""
    g = OrbmGrid()
    g.minX, g.minY = minX, minY
    g.invW = np.float32(cols) / np.float32(np.float32(maxX) - np.float32(minX))
    g.invH = np.float32(rows) / np.float32(np.float32(maxY) - np.float32(minY))
    g.cols, g.rows = cols, rows
    return g


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # ORBmatcher.cc:37-39

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        """ORBmatcher(float nnratio=0.6, bool checkOri=true) -- ORBmatcher.h:41"""
        self.mfNNratio, self.mbCheckOrientation = float(nnratio), bool(checkOri)
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.orbm_create(int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # static int DescriptorDistance(const cv::Mat&, const cv::Mat&) -- ORBmatcher.cc:1649
    def DescriptorDistance(self, a, b):
        return int(self.distance_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])

    def distance_matrix(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        out = np.zeros((q.shape[0], t.shape[0]), dtype=np.int32)
        check(self._L.orbm_distance_matrix(self._h, ptr(q), q.shape[0], ptr(t), t.shape[0], ptr(out)))
        return out

    def match_bruteforce(self, qdesc, qangle, tdesc, tangle, th_low=50):
        qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
        tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
        qangle = np.ascontiguousarray(qangle, dtype=np.float32)
        tangle = np.ascontiguousarray(tangle, dtype=np.float32)
        nq, nt = qdesc.shape[0], tdesc.shape[0]
        match = np.full(max(nq, 1), -1, dtype=np.int32)
        n = C.c_int(0)
        check(self._L.orbm_match_bruteforce(self._h, ptr(qdesc), ptr(qangle), nq, ptr(tdesc), ptr(tangle), nt,
                                            C.c_float(self.mfNNratio), int(th_low), int(self.mbCheckOrientation),
                                            ptr(match), C.byref(n)))
        return match[:nq], n.value

    @staticmethod
    def _fv(node_id, start, idx):
        fv = OrbmFeatVec()
        keep = (np.ascontiguousarray(node_id, dtype=np.uint32), np.ascontiguousarray(start, dtype=np.int32),
                np.ascontiguousarray(idx, dtype=np.int32))
        fv.n_nodes = keep[0].shape[0]
        fv.node_id, fv.start, fv.idx = (k.ctypes.data for k in keep)
        return fv, keep

    def SearchByBoW(self, qdesc, qangle, qvalid, qfv, tdesc, tangle, tvalid, tfv, out_by_train=True):
        """SearchByBoW(KeyFrame*, Frame&, ...) (out_by_train=True, ORBmatcher.cc:159) or
        SearchByBoW(KeyFrame*, KeyFrame*, ...) (out_by_train=False, :524).  qfv/tfv = (node_id, start, idx)."""
        qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
        tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
        qangle = np.ascontiguousarray(qangle, dtype=np.float32)
        tangle = np.ascontiguousarray(tangle, dtype=np.float32)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        tv = None if tvalid is None else np.ascontiguousarray(tvalid, dtype=np.uint8)
        nq, nt = qdesc.shape[0], tdesc.shape[0]
        fq, kq = self._fv(*qfv)
        ft, kt = self._fv(*tfv)
        nout = nt if out_by_train else nq
        match = np.full(max(nout, 1), -1, dtype=np.int32)
        n = C.c_int(0)
        check(self._L.orbm_search_by_bow(self._h, ptr(qdesc), ptr(qangle), ptr(qv), nq, C.byref(fq),
                                         ptr(tdesc), ptr(tangle), ptr(tv), nt, C.byref(ft),
                                         C.c_float(self.mfNNratio), int(self.mbCheckOrientation), int(bool(out_by_train)),
                                         ptr(match), C.byref(n)))
        return match[:nout], n.value

    def SearchByProjection(self, mode, th_dist, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos,
                           grid, t_keys_un, tdesc, t_occ, assign):
        """The four SearchByProjection overloads, flattened (mode 3/4/5/6, see orbslamm_hip.h)."""
        pp = OrbmProjParams(int(mode), self.mfNNratio, int(self.mbCheckOrientation), int(th_dist))
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
        n = C.c_int(0)
        check(self._L.orbm_search_by_projection(self._h, C.byref(pp), ptr(q_uvr), ptr(q_lvl), ptr(qdesc), ptr(qangle),
                                                ptr(qv), ptr(qo), q_uvr.shape[0], C.byref(grid), ptr(t_keys_un),
                                                ptr(tdesc), t_keys_un.shape[0], ptr(t_occ), ptr(assign), C.byref(n)))
        return assign, t_occ, n.value

    def ProjectionPrepare(self, grid, t_keys_un, tdesc):
        """orbm_projection_prepare: the train side of the next SearchByProjection goes up now (pass the SAME contiguous arrays to
        that call: the hint is matched by pointer)"""
        assert t_keys_un.flags["C_CONTIGUOUS"] and t_keys_un.dtype == KP_DTYPE and tdesc.flags["C_CONTIGUOUS"] and tdesc.dtype == np.uint8
        check(self._L.orbm_projection_prepare(self._h, C.byref(grid), ptr(t_keys_un), ptr(tdesc), t_keys_un.shape[0]))

    # ---- the Frame's matcher-side state in HBM
    def frame_from_device(self, d_keys, d_desc, n, K, D, grid):
        """Frame::Frame tail (UndistortKeyPoints + AssignFeaturesToGrid) on device-resident extractor output;
        d_keys / d_desc are device addresses (orbx_device_results + frame offsets).  Returns an opaque frame."""
        K = np.ascontiguousarray(K, dtype=np.float32)
        D = np.ascontiguousarray(D, dtype=np.float32)
        f = C.c_void_p()
        check(self._L.orbm_frame_create(self._h, C.c_void_p(int(d_keys)), C.c_void_p(int(d_desc)), int(n), ptr(K), ptr(D),
                                        C.byref(grid), C.byref(f)))
        return f

    def frame_keys_un(self, frame):
        n = self._L.orbm_frame_size(frame)
        out = np.zeros(n, dtype=KP_DTYPE)
        check(self._L.orbm_frame_download_keys_un(frame, ptr(out)))
        return out

    def frame_settle(self, frame):
        """orbm_frame_settle: returns when the frame's build has read the device arrays it was made from"""
        check(self._L.orbm_frame_settle(frame))

    def frame_destroy(self, frame):
        check(self._L.orbm_frame_destroy(frame))

    def frame_compute_bow(self, frame, voc, levelsup=4):
        """Frame::ComputeBoW on a device-resident frame: returns the BowVector (word ids, values); the
        FeatureVector stays with the frame (SearchByBoWFrames)."""
        n = self._L.orbm_frame_size(frame)
        wid = np.zeros(max(n, 1), dtype=np.uint32)
        wval = np.zeros(max(n, 1), dtype=np.float64)
        nw = C.c_int(0)
        check(self._L.orbm_frame_compute_bow(frame, voc._h, int(levelsup), ptr(wid), ptr(wval), C.byref(nw)))
        return wid[:nw.value].copy(), wval[:nw.value].copy()

    def SearchByBoWFrames(self, qframe, qvalid, tframe, tvalid, out_by_train=True):
        nout = self._L.orbm_frame_size(tframe if out_by_train else qframe)
        match = np.full(max(nout, 1), -1, dtype=np.int32)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        tv = None if tvalid is None else np.ascontiguousarray(tvalid, dtype=np.uint8)
        n = C.c_int(0)
        check(self._L.orbm_search_by_bow_frames(self._h, qframe, ptr(qv), tframe, ptr(tv), C.c_float(self.mfNNratio),
                                                int(self.mbCheckOrientation), int(bool(out_by_train)), ptr(match), C.byref(n)))
        return match[:nout], n.value

    def SearchForInitializationFrames(self, q_xy, window_size, f1, f2):
        q_xy = np.ascontiguousarray(q_xy, dtype=np.float32)
        n1 = self._L.orbm_frame_size(f1)
        m12 = np.full(max(n1, 1), -1, dtype=np.int32)
        n = C.c_int(0)
        check(self._L.orbm_search_for_initialization_frames(self._h, ptr(q_xy), C.c_float(window_size), f1, f2, C.c_float(self.mfNNratio),
                                                            int(self.mbCheckOrientation), ptr(m12), C.byref(n)))
        return m12[:n1], n.value

    def SearchByProjectionFrame(self, mode, th_dist, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, frame, t_occ, assign):
        """SearchByProjection with a device-resident frame as train side."""
        pp = OrbmProjParams(int(mode), self.mfNNratio, int(self.mbCheckOrientation), int(th_dist))
        q_uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
        q_lvl = np.ascontiguousarray(q_lvl, dtype=np.int8)
        qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
        qangle = np.ascontiguousarray(qangle, dtype=np.float32)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        qo = None if q_obs_pos is None else np.ascontiguousarray(q_obs_pos, dtype=np.uint8)
        t_occ = np.ascontiguousarray(t_occ, dtype=np.uint8).copy()
        assign = np.ascontiguousarray(assign, dtype=np.int32).copy()
        n = C.c_int(0)
        check(self._L.orbm_search_by_projection_frame(self._h, C.byref(pp), ptr(q_uvr), ptr(q_lvl), ptr(qdesc), ptr(qangle),
                                                      ptr(qv), ptr(qo), q_uvr.shape[0], frame, ptr(t_occ), ptr(assign), C.byref(n)))
        return assign, t_occ, n.value

    def last_search_stats(self):
        """(rounds, candidates) of this handle's last projection search"""
        r, c = C.c_int(0), C.c_int(0)
        check(self._L.orbm_last_search_stats(self._h, C.byref(r), C.byref(c)))
        return r.value, c.value

    def frame_set(self, slots, cap, K, D, grid, bounds, scale_factors):
        """a set of device-resident frames (FrameSet): Frame::Frame's tail and SearchByProjection(Cur, Last) batched"""
        return FrameSet(self, slots, cap, K, D, grid, bounds, scale_factors)

    def window_best_frame(self, q_uvr, q_pred, qdesc, qvalid, frame, inv_sigma2=None, chi2=False):
        """orbm_window_best with a device-resident frame as train side (the KeyFrame of Fuse / SearchBySim3)"""
        q_uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
        q_pred = np.ascontiguousarray(q_pred, dtype=np.int8)
        qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        sig = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, dtype=np.float32)
        nq = q_uvr.shape[0]
        bi = np.full(max(nq, 1), -1, np.int32)
        bd = np.full(max(nq, 1), 256, np.int32)
        check(self._L.orbm_window_best_frame(self._h, ptr(q_uvr), ptr(q_pred), ptr(qdesc), ptr(qv), nq, frame, ptr(sig),
                                             0 if sig is None else sig.shape[0], int(bool(chi2)), ptr(bi), ptr(bd)))
        return bi[:nq], bd[:nq]

    def SearchForTriangulationFrames(self, f1, skip1, f2, skip2, F12, ex, ey, sf2, sigma2_2):
        """SearchForTriangulation between two device-resident frames that ran frame_compute_bow"""
        s1 = None if skip1 is None else np.ascontiguousarray(skip1, dtype=np.uint8)
        s2 = None if skip2 is None else np.ascontiguousarray(skip2, dtype=np.uint8)
        F = np.ascontiguousarray(F12, dtype=np.float32).reshape(9)
        sf2 = np.ascontiguousarray(sf2, dtype=np.float32)
        sg2 = np.ascontiguousarray(sigma2_2, dtype=np.float32)
        n1 = self._L.orbm_frame_size(f1)
        m12 = np.full(max(n1, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbm_search_for_triangulation_frames(self._h, f1, ptr(s1), f2, ptr(s2), ptr(F), C.c_float(ex), C.c_float(ey),
                                                           ptr(sf2), ptr(sg2), sf2.shape[0], int(self.mbCheckOrientation), ptr(m12), C.byref(n)))
        return m12[:n1], n.value

    # ---- SURVEY.md 8(f) rank 1
    def window_best(self, q_uvr, q_pred, qdesc, qvalid, grid, t_keys_un, tdesc, inv_sigma2=None, chi2=False,
                    q_ur=None, t_uright=None):
        """device part of Fuse (ORBmatcher.cc:827, chi2=True), Fuse(KF,Scw) (:977) and SearchBySim3 (:1104)"""
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
        check(self._L.orbm_window_best(self._h, ptr(q_uvr), ptr(qur), ptr(q_pred), ptr(qdesc), ptr(qv), nq, C.byref(grid),
                                       ptr(t_keys_un), ptr(tdesc), ptr(tur), t_keys_un.shape[0], ptr(sig),
                                       0 if sig is None else sig.shape[0], int(bool(chi2)), ptr(bi), ptr(bd)))
        return bi[:nq], bd[:nq]

    def SearchForInitialization(self, q_xy, window_size, q_keys_un, qdesc, grid, t_keys_un, tdesc):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) -- ORBmatcher.cc:407"""
        q_xy = np.ascontiguousarray(q_xy, dtype=np.float32)
        q_keys_un = np.ascontiguousarray(q_keys_un, dtype=KP_DTYPE)
        qdesc = np.ascontiguousarray(qdesc, dtype=np.uint8)
        t_keys_un = np.ascontiguousarray(t_keys_un, dtype=KP_DTYPE)
        tdesc = np.ascontiguousarray(tdesc, dtype=np.uint8)
        nq = q_keys_un.shape[0]
        m12 = np.full(max(nq, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbm_search_for_initialization(self._h, ptr(q_xy), C.c_float(window_size), ptr(q_keys_un), ptr(qdesc), nq,
                                                     C.byref(grid), ptr(t_keys_un), ptr(tdesc), t_keys_un.shape[0],
                                                     C.c_float(self.mfNNratio), int(self.mbCheckOrientation), ptr(m12), C.byref(n)))
        return m12[:nq], n.value

    def SearchForTriangulation(self, k1, d1, skip1, fv1, k2, d2, skip2, fv2, F12, ex, ey, sf2, sigma2_2,
                               only_stereo=False, uright1=None, uright2=None):
        """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) -- ORBmatcher.cc:659"""
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
        f1, keep1 = self._fv(*fv1)
        f2, keep2 = self._fv(*fv2)
        n1 = k1.shape[0]
        m12 = np.full(max(n1, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbm_search_for_triangulation(self._h, ptr(k1), ptr(d1), ptr(s1), ptr(u1), n1, C.byref(f1),
                                                    ptr(k2), ptr(d2), ptr(s2), ptr(u2), k2.shape[0], C.byref(f2),
                                                    ptr(F), C.c_float(ex), C.c_float(ey), ptr(sf2), ptr(sg2), sf2.shape[0],
                                                    int(bool(only_stereo)), int(self.mbCheckOrientation), ptr(m12), C.byref(n)))
        return m12[:n1], n.value

    def UndistortKeyPoints(self, keys, K, D):
        """Frame::UndistortKeyPoints (Frame.cc:404): K = (fx, fy, cx, cy), D = (k1, k2, p1, p2, k3)"""
        keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
        K = np.ascontiguousarray(K, dtype=np.float32)
        D = np.ascontiguousarray(D, dtype=np.float32)
        out = np.zeros_like(keys)
        check(self._L.orbm_undistort_keypoints(self._h, ptr(keys), keys.shape[0], ptr(K), ptr(D), ptr(out)))
        return out

    def ComputeStereoFromRGBD(self, keys, keys_un, depth, mbf):
        """Frame::ComputeStereoFromRGBD (Frame.cc:641-663): depth = float32 image; returns (mvuRight, mvDepth)"""
        keys = np.ascontiguousarray(keys, dtype=KP_DTYPE)
        keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        n = keys.shape[0]
        ur = np.zeros(n, np.float32); dp = np.zeros(n, np.float32)
        check(self._L.orbm_compute_stereo_from_rgbd(self._h, ptr(keys), ptr(keys_un), n, ptr(depth), depth.shape[1], depth.shape[0], depth.shape[1],
                                                    C.c_float(mbf), ptr(ur), ptr(dp)))
        return ur, dp

    def ComputeDistinctiveDescriptors(self, desc, start):
        """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242), batched: desc = all observations'
        descriptors back to back, start = CSR offsets per map point; returns the winning index per point"""
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        start = np.ascontiguousarray(start, dtype=np.int32)
        npts = start.shape[0] - 1
        out = np.full(max(npts, 1), -1, np.int32)
        check(self._L.orbm_distinctive_descriptors(self._h, ptr(desc), ptr(start), npts, ptr(out)))
        return out[:npts]

    def GetFeaturesInArea(self, grid, keys_un, x, y, r, minLevel=-1, maxLevel=-1):
        """Frame::GetFeaturesInArea (Frame.cc:327-380) evaluated on the device grid"""
        keys_un = np.ascontiguousarray(keys_un, dtype=KP_DTYPE)
        out = np.zeros(max(keys_un.shape[0], 1), dtype=np.int32)
        n = C.c_int(0)
        check(self._L.orbm_features_in_area(self._h, C.byref(grid), ptr(keys_un), keys_un.shape[0], C.c_float(x),
                                            C.c_float(y), C.c_float(r), int(minLevel), int(maxLevel), ptr(out),
                                            out.shape[0], C.byref(n)))
        return out[:n.value].copy()


class FrameSet:
    """`slots` device-resident frames (mvKeysUn, descriptors, mGrid) and the frame-to-frame projection search over
    pairs of them, one launch per batch (include/orbslamm_hip.h: orbm_frameset_*, orbm_track_*)."""

    def __init__(self, matcher, slots, cap, K, D, grid, bounds, scale_factors):
        self._m = matcher
        self._L = matcher._L
        self.slots, self.cap = int(slots), int(cap)
        K = np.ascontiguousarray(K, dtype=np.float32)
        D = np.ascontiguousarray(D, dtype=np.float32)
        b = np.ascontiguousarray(bounds, dtype=np.float32)
        sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self._h = C.c_void_p()
        check(self._L.orbm_frameset_create(matcher._h, self.slots, self.cap, ptr(K), ptr(D), C.byref(grid), ptr(b), ptr(sf), sf.shape[0],
                                           C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbm_frameset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, slot0, n, d_keys, d_desc, d_counts, src_cap):
        check(self._L.orbm_frameset_build(self._h, int(slot0), int(n), C.c_void_p(int(d_keys)), C.c_void_p(int(d_desc)),
                                          C.c_void_p(int(d_counts)), int(src_cap)))

    def build_from_extractor(self, slot0, extractor):
        check(self._L.orbm_frameset_build_from_extractor(self._h, int(slot0), extractor._h))

    def sync(self):
        check(self._L.orbm_frameset_sync(self._h))

    def attach(self, extractor):
        """orbm_frameset_attach: the set's kernels ride on the extractor's stream (the live, one-frame-per-call chain);
        None detaches"""
        check(self._L.orbm_frameset_attach(self._h, extractor._h if extractor is not None else None))

    def download(self, slot):
        keys = np.zeros(self.cap, dtype=KP_DTYPE)
        desc = np.zeros((self.cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        check(self._L.orbm_frameset_download(self._h, int(slot), ptr(keys), ptr(desc), self.cap, C.byref(n)))
        return keys[:n.value].copy(), desc[:n.value].copy()

    def track(self, cur_slots, last_slots, th=15.0, th_dist=100, nnratio=0.9, check_ori=True):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono=true) for every (cur, last) slot pair; asynchronous"""
        pp = OrbmProjParams(4, float(nnratio), int(bool(check_ori)), int(th_dist))
        cs = np.ascontiguousarray(cur_slots, dtype=np.int32)
        ls = np.ascontiguousarray(last_slots, dtype=np.int32)
        self._npairs = cs.shape[0]
        check(self._L.orbm_track_frames(self._h, C.byref(pp), C.c_float(th), ptr(cs), ptr(ls), cs.shape[0]))

    def track_local_points(self, slot, q_uvr, q_lvl, qdesc, qvalid=None, q_obs_pos=None, t_occ=None, th_dist=100, nnratio=0.8, mode=3):
        """orbm_track_local_points: SearchByProjection(Frame, local MapPoints, th) against the frame in `slot`; asynchronous,
        the table comes back through results() as one pair"""
        pp = OrbmProjParams(int(mode), float(nnratio), 0, int(th_dist))
        uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
        lvl = np.ascontiguousarray(q_lvl, dtype=np.int8)
        qd = np.ascontiguousarray(qdesc, dtype=np.uint8)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        qo = None if q_obs_pos is None else np.ascontiguousarray(q_obs_pos, dtype=np.uint8)
        occ = None
        if t_occ is not None:
            occ = np.zeros(self.cap, np.uint8)
            occ[:len(t_occ)] = t_occ
        check(self._L.orbm_track_local_points(self._h, int(slot), C.byref(pp), ptr(uvr), ptr(lvl), ptr(qd), ptr(qv), ptr(qo), uvr.shape[0], ptr(occ)))

    def track_projected(self, cur_slot, last_slot, q_uvr, q_lvl, qvalid=None, q_obs_pos=None, t_occ=None, th_dist=100, nnratio=0.9,
                        check_ori=True, mode=4):
        """orbm_track_frame_projected: SearchByProjection(Cur, Last) with the caller's projections of LastFrame's features"""
        pp = OrbmProjParams(int(mode), float(nnratio), int(bool(check_ori)), int(th_dist))
        uvr = np.ascontiguousarray(q_uvr, dtype=np.float32)
        lvl = np.ascontiguousarray(q_lvl, dtype=np.int8)
        qv = None if qvalid is None else np.ascontiguousarray(qvalid, dtype=np.uint8)
        qo = None if q_obs_pos is None else np.ascontiguousarray(q_obs_pos, dtype=np.uint8)
        occ = None
        if t_occ is not None:
            occ = np.zeros(self.cap, np.uint8)
            occ[:len(t_occ)] = t_occ
        check(self._L.orbm_track_frame_projected(self._h, int(cur_slot), int(last_slot), C.byref(pp), ptr(uvr), ptr(lvl), ptr(qv), ptr(qo), uvr.shape[0], ptr(occ)))

    def results(self, back=0):
        """waits for the last (back=0) or last-but-one (back=1) track(); (assign[npairs][cap] view of the pinned result
        block, nmatches[npairs])"""
        a, n, cap, npairs = C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_int(0)
        check(self._L.orbm_track_results(self._h, int(back), C.byref(a), C.byref(n), C.byref(npairs), C.byref(cap)))
        np_ = npairs.value
        if np_ == 0:
            return np.zeros((0, self.cap), np.int32), np.zeros(0, np.int32)
        assign = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_int32)), shape=(np_, cap.value))
        nm = np.ctypeslib.as_array(C.cast(n, C.POINTER(C.c_int32)), shape=(np_,))
        return assign, nm

    def stats(self, pair):
        r, c = C.c_int(0), C.c_int(0)
        check(self._L.orbm_track_stats(self._h, int(pair), C.byref(r), C.byref(c)))
        return r.value, c.value

    # ---- BoW side: Frame::ComputeBoW per slot, SearchByBoW(KeyFrame, Frame) per slot pair
    def compute_bow(self, voc, slot0, n, levelsup=4):
        check(self._L.orbm_frameset_compute_bow(self._h, voc._h, int(slot0), int(n), int(levelsup)))

    def bow_vector(self, slot):
        wid = np.zeros(self.cap, np.uint32)
        wv = np.zeros(self.cap, np.float64)
        n = C.c_int(0)
        check(self._L.orbm_frameset_bow_vector(self._h, int(slot), ptr(wid), ptr(wv), self.cap, C.byref(n)))
        return wid[:n.value].copy(), wv[:n.value].copy()

    def search_by_bow(self, kf_slots, frame_slots, nnratio=0.7, check_ori=True):
        ks = np.ascontiguousarray(kf_slots, dtype=np.int32)
        fs = np.ascontiguousarray(frame_slots, dtype=np.int32)
        check(self._L.orbm_bow_frames(self._h, ptr(ks), ptr(fs), ks.shape[0], C.c_float(nnratio), int(bool(check_ori))))

    def bow_results(self, back=0):
        a, n, cap, npairs = C.c_void_p(), C.c_void_p(), C.c_int(0), C.c_int(0)
        check(self._L.orbm_bow_results(self._h, int(back), C.byref(a), C.byref(n), C.byref(npairs), C.byref(cap)))
        if npairs.value == 0:
            return np.zeros((0, self.cap), np.int32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_int32)), shape=(npairs.value, cap.value)),
                np.ctypeslib.as_array(C.cast(n, C.POINTER(C.c_int32)), shape=(npairs.value,)))


def descriptors_to_text(desc):
    """MapSerializer's descriptor attribute (src/MapSerializer.cc:344-347, 429-431): `os << cv::Mat` of a CV_8U matrix"""
    L = lib()
    d = np.ascontiguousarray(desc, dtype=np.uint8)
    if d.ndim == 1:
        d = d[None]
    n, cols = d.shape
    ln = C.c_size_t(0)
    check(L.orbm_descriptors_to_text(ptr(d), n, cols, None, C.c_size_t(0), C.byref(ln)))
    buf = C.create_string_buffer(ln.value + 1)
    check(L.orbm_descriptors_to_text(ptr(d), n, cols, buf, C.c_size_t(ln.value + 1), C.byref(ln)))
    return buf.value.decode()


def descriptors_from_text(text, cols=32):
    L = lib()
    n = C.c_int(0)
    t = text.encode()
    check(L.orbm_descriptors_from_text(t, None, 0, cols, C.byref(n)))
    out = np.zeros((n.value, cols), dtype=np.uint8)
    check(L.orbm_descriptors_from_text(t, ptr(out), n.value, cols, C.byref(n)))
    return out
