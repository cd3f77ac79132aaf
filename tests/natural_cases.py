"""The natural-image fixture (tests/golden/natural.npz, made by tests/golden/make_natural.py): every frame the tests
extract from it, by name."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "natural.npz")


def load():
    """name -> (uint8 image, nfeatures), and name -> "n:sha256(kps):sha256(desc)" as the oracle produced them"""
    z = np.load(PATH)
    frames = {}
    for k in sorted(z.files):
        if k.startswith("c2_"):
            frames[k] = (z[k], 1000)
    cv = z["c3_canvas"]
    for t in range(3):  # a camera panning over the retina photograph: origin (2t, t)
        frames["c3_pan%d" % t] = (np.ascontiguousarray(cv[t:t + 376, 2 * t:2 * t + 1241]), 2000)
    frames["c3_mosaic"] = (z["c3_mosaic"], 2000)
    frames["c3_hubble"] = (z["c3_hubble"], 2000)
    frames["stereo_l"] = (z["stereo_l"], 1200)
    frames["stereo_r"] = (z["stereo_r"], 1200)
    digests = dict(zip([str(n) for n in z["digest_names"]], [str(v) for v in z["digest_values"]]))
    return frames, digests
