#!/usr/bin/env python3
"""FAST-9/16 segment-test fixture from an INDEPENDENT published implementation:
scikit-image's skimage.feature.corner_fast (run with the conda python of the build
container: /opt/conda/bin/python tests/golden/make_fast_detect_fixture.py).

What it pins: the corner PREDICATE of FAST-9 (>= 9 contiguous ring pixels all brighter than
v+t or all darker than v-t, strict) on a seeded image, at the two thresholds of the path (20, 7).
What it does not pin: OpenCV's corner score, its 3x3 non-max suppression and border handling
(those stay "parity unpinned").  The image is passed as float64 holding the
integer grey values so that skimage's comparisons are exact."""
import os

import numpy as np
from skimage.feature import corner_fast

HERE = os.path.dirname(os.path.abspath(__file__))


def image():
    rng = np.random.RandomState(12345)
    img = np.full((96, 128), 120, np.int32)
    for _ in range(60):
        x, y, w, h, v = rng.randint(0, 128), rng.randint(0, 96), rng.randint(3, 30), rng.randint(3, 30), rng.randint(0, 256)
        img[y:y + h, x:x + w] = v
    img = np.clip(img + rng.randint(-6, 7, img.shape), 0, 255)
    return img.astype(np.uint8)


def main():
    img = image()
    out = {"image": img}
    for t in (20, 7):
        resp = corner_fast(img.astype(np.float64), n=9, threshold=float(t))
        out["corners_t%d" % t] = (resp > 0).astype(np.uint8)
        print("t=%d: %d corner pixels" % (t, int((resp > 0).sum())))
    np.savez_compressed(os.path.join(HERE, "fast_detect_skimage.npz"), **out)


if __name__ == "__main__":
    main()
