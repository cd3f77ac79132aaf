"""The oracle on natural images (skimage.data photographs and textures at the benchmark shapes): its keypoints and
descriptors still hash to what tests/golden/make_natural.py recorded.  A regression pin of the ORACLE (the digests are
its own output, not the reference's: parity stays unpinned at the OpenCV boundary) and the proof
that the GPU suite's natural-image inputs are the committed bytes."""
import hashlib

import numpy as np

from natural_cases import load


def _d(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_oracle_reproduces_the_natural_image_digests(oracle):
    frames, digests = load()
    assert len(frames) == 13 and set(frames) == set(digests)
    for name in ("c2_camera", "c2_coffee", "c3_pan1", "c3_mosaic", "stereo_l"):   # a sample: the whole set takes ~1 s per frame
        img, nf = frames[name]
        r = oracle.Extractor(nf, 1.2, 8, 20, 7)(img)
        assert "%d:%s:%s" % (len(r["kps"]), _d(r["kps"]), _d(r["desc"])) == digests[name], name


def test_natural_images_are_not_synthetic_looking():
    """what the fixture is for: the synthetic scene is flat regions (+-4 noise) cut by step edges, so hardly any
    neighbouring pixels differ by a MODERATE amount; photographs and textures are full of such gradients"""
    frames, _ = load()
    from orbslamm_amd import synth

    def moderate(img):
        d = np.abs(np.diff(img.astype(np.int32), axis=1))
        return np.mean((d > 8) & (d <= 40))

    syn = moderate(synth.make_frames(640, 480, 1)[0])
    assert syn < 0.03
    for name in ("c2_camera", "c2_grass", "c2_brick", "c2_coffee", "c3_mosaic", "c3_hubble", "stereo_l"):
        assert moderate(frames[name][0]) > 3 * syn, name
