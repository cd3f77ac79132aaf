#!/usr/bin/env python3
"""bench.py's tracking_path block alone (quick iteration on the Tracking-shaped matchers); prints its JSON."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from orbslamm_amd import ORBextractor, synth  # noqa: E402

cfg = bench.CONFIGS[os.environ.get("TRK_CONFIG", "c3")]
B = int(os.environ.get("TRK_BATCH", "64"))
ex = ORBextractor(cfg["nfeat"], 1.2, 8, 20, 7, max_width=cfg["w"], max_height=cfg["h"], max_batch=B, device=0)
canvas = synth.make_scene(cfg["w"], cfg["h"], 0)
dargs, first = [], None
for p in range(2):
    fr = np.stack([synth.frame_from_scene(canvas, cfg["w"], cfg["h"], p * B + t, 0) for t in range(B)])
    first = fr if p == 0 else first
    dargs.append(ex.upload_frames(fr, stride=cfg["stride"]))
print(json.dumps(bench.tracking_path(ex, cfg, first, dargs, seconds=float(os.environ.get("TRK_SECONDS", "1.0")))))
