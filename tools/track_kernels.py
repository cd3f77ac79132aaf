#!/usr/bin/env python3
"""The frame-set kernels alone, for rocprofv3 --kernel-trace --stats: B frames built and B pairs searched per step on one
extracted batch (nothing else running).  usage: track_kernels.py B [steps] [th]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
th = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
W, H, NF = 1241, 376, 2000
fr = synth.make_frames(W, H, max(B, 2), stream=0)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=max(B, 2), device=0)
m = ORBmatcher(0.9, True, device=0)
sf = np.array(ex.GetScaleFactors(), np.float32)
g = make_grid(0.0, 0.0, float(W), float(H))
fs = m.frame_set(max(B, 2), ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0] * 5, g, [0.0, float(W), 0.0, float(H)], sf)
ex.extract_batch_device(*ex.upload_frames(fr))
ex.sync()
voc = synth.make_vocabulary(10, 6)
G = ORBVocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
cur = np.arange(B) if B > 1 else np.array([1])
last = np.roll(cur, 1) if B > 1 else np.array([0])
for s in range(steps):
    fs.build_from_extractor(0, ex)
    fs.track(cur[:B], last[:B], th=th)
    fs.results()
    fs.compute_bow(G, 0, max(B, 2), 4)
    fs.search_by_bow(last[:B], cur[:B], 0.7, True)
    fs.bow_results()
