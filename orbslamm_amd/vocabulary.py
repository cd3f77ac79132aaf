"""Host-side mirror of ORBVocabulary (include/ORBVocabulary.h = DBoW2 TemplatedVocabulary<FORB>)
for the one member the per-frame path uses: transform() as called by Frame::ComputeBoW
(src/Frame.cc:395-402).  The tree lives in HBM; the descent and the BowVector / FeatureVector
aggregation are HIP kernels."""
import ctypes as C

import numpy as np

from ._lib import check, lib, ptr

L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)   # DBoW2 ScoringType
TF_IDF, TF, IDF, BINARY = range(4)                                         # DBoW2 WeightingType


class ORBVocabulary:
    def __init__(self, k=None, L=None, scoring=L1_NORM, weighting=TF_IDF, parent=None, is_leaf=None, desc=None,
                 weight=None, device=0, _handle=None):
        self._L = lib()
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
            return
        parent = np.ascontiguousarray(parent, dtype=np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, dtype=np.uint8)
        desc = np.ascontiguousarray(desc, dtype=np.uint8)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        check(self._L.orbv_create(int(device), int(k), int(L), int(scoring), int(weighting), parent.shape[0], ptr(parent),
                                  ptr(is_leaf), ptr(desc), ptr(weight), C.byref(self._h)))

    @classmethod
    def loadFromTextFile(cls, path, device=0):
        """bool loadFromTextFile(const std::string&) -- TemplatedVocabulary.h:1338 (ORBvoc.txt format)"""
        L = lib()
        h = C.c_void_p()
        check(L.orbv_load_text(int(device), str(path).encode(), C.byref(h)))
        return cls(_handle=h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, descriptors, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup) -- TemplatedVocabulary.h:1127.
        returns (word_ids, values), (node_ids, start, feature_idx)"""
        d = np.ascontiguousarray(descriptors, dtype=np.uint8)
        n = d.shape[0]
        wid = np.zeros(max(n, 1), np.uint32)
        wv = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32)
        fs = np.zeros(n + 1, np.int32)
        fi = np.zeros(max(n, 1), np.int32)
        nw, nf = C.c_int(), C.c_int()
        check(self._L.orbv_transform(self._h, ptr(d), n, int(levelsup), ptr(wid), ptr(wv), C.byref(nw), ptr(fn), ptr(fs), ptr(fi),
                                     C.byref(nf)))
        return (wid[:nw.value].copy(), wv[:nw.value].copy()), (fn[:nf.value].copy(), fs[:nf.value + 1].copy(), fi[:fs[nf.value]].copy())
