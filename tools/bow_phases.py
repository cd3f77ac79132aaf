#!/usr/bin/env python3
"""Phase times of one k_voc_aggregate_set workgroup (library built with -DORBT_PHASE_TIMING, see tools/proj_phases.sh)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth, _lib  # noqa: E402

W, H, NF = 1241, 376, 2000
fr = synth.make_frames(W, H, 2, stream=0)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, device=0)
m = ORBmatcher(0.7, True, device=0)
sf = np.array(ex.GetScaleFactors(), np.float32)
g = make_grid(0.0, 0.0, float(W), float(H))
fs = m.frame_set(2, ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0] * 5, g, [0.0, float(W), 0.0, float(H)], sf)
ex.extract_batch_device(*ex.upload_frames(fr))
fs.build_from_extractor(0, ex)
voc = synth.make_vocabulary(10, 6)
G = ORBVocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
L = _lib.lib()
names = ["sort words", "boundaries + sums", "norm", "node keys", "sort nodes", "FeatureVector out"]
acc = np.zeros(6)
for rep in range(20):
    fs.compute_bow(G, 0, 2, 4)
    fs.sync()
    us = (C.c_double * 6)()
    L.orbv_debug_phase_times(us, 6)
    if rep >= 5:
        acc += np.array(list(us))
acc /= 15
print(", ".join("%s %.1f us" % (k, v) for k, v in zip(names, acc)) + " | total %.1f us" % acc.sum())
