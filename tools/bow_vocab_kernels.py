#!/usr/bin/env python3
"""ComputeBoW + SearchByBoW(KF, F) on one frame pair, 40 times per vocabulary shape (complete / ORBvoc-sized ragged / strongly
ragged): run under `rocprofv3 --kernel-trace --stats` to see which kernel pays for the shape (tools/bow_vocab_kernels.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslamm_amd import ORBextractor, ORBmatcher, ORBVocabulary, make_grid, synth  # noqa: E402

W, H, NF = 1241, 376, 2000
which = sys.argv[1] if len(sys.argv) > 1 else "complete"
rg = {"complete": False, "orbvoc": 0.994, "strong": True}[which]
fr = synth.make_frames(W, H, 2, stream=0)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, device=0)
m = ORBmatcher(0.7, True, device=0)
sf = np.array(ex.GetScaleFactors(), np.float32)
g = make_grid(0.0, 0.0, float(W), float(H))
fs = m.frame_set(2, ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0] * 5, g, [0.0, float(W), 0.0, float(H)], sf)
ex.extract_batch_device(*ex.upload_frames(fr))
fs.build_from_extractor(0, ex)
voc = synth.make_vocabulary(10, 6, ragged=rg)
G = ORBVocabulary(10, 6, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"], device=0)
ts = []
for i in range(40):
    t0 = time.perf_counter()
    fs.compute_bow(G, 0, 2, 4)
    fs.search_by_bow([0], [1], 0.7, True)
    mm, nn = fs.bow_results()
    ts.append(time.perf_counter() - t0)
print(which, "nodes", len(voc["parent"]), "median ms", float(np.median(ts[5:])) * 1e3, "matches", int(nn[0]))
