#!/opt/conda/bin/python3.9
"""Natural-image parity inputs (round 3).  Run in the BUILD container with the interpreter that has scikit-image:

    /opt/conda/bin/python3.9 tests/golden/make_natural.py

Writes tests/golden/natural.npz: grey photographs and textures from skimage.data (the reference's own inputs are
TUM/KITTI photographs, Examples/Monocular/mono_tum.cc:60-78 -- not in the container; these at least have natural
statistics, which the rectangles-and-discs scenes of orbslamm_amd/synth.py do not) at the benchmark shapes, and for each
the SHA-256 of what the CPU oracle extracts from it.  The digests are ORACLE-generated regression vectors (provenance:
oracle/orb_extract.c at the commit that wrote them), not reference outputs: they pin the oracle against drift, the GPU
tests compare the HIP path with the oracle live on the same bytes.

  c2_*        640x480  (BASELINE.json configs[1]): six images resized with skimage.transform.resize(order=1, anti_aliasing)
  c3_canvas   392x1305 crop of the retina photograph at native resolution: the tests cut the 1241x376 frames of a panning
              camera out of it (origin (2t, t)), so consecutive frames match like a real sequence
  c3_mosaic   1241x376: three native-resolution crops side by side (camera | astronaut | gravel)
  c3_hubble   1241x376: crop of the Hubble deep field resized x1.241
  stereo_l/r  741x500: the Middlebury motorcycle pair (rectified), grey -- Frame::ComputeStereoMatches' kind of input
"""
import hashlib
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def grey(im):
    from skimage.color import rgb2gray
    if im.ndim == 3:
        im = np.round(rgb2gray(im[..., :3]) * 255.0)
    return np.clip(im, 0, 255).astype(np.uint8)


def resized(im, h, w):
    from skimage.transform import resize
    out = resize(grey(im).astype(np.float64), (h, w), order=1, anti_aliasing=True, preserve_range=True, mode="reflect")
    return np.clip(np.round(out), 0, 255).astype(np.uint8)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def frames_of(z):
    """name -> (image, nfeatures): every frame the tests extract from the fixture"""
    out = {}
    for k in sorted(z):
        if k.startswith("c2_"):
            out[k] = (z[k], 1000)
    cv = z["c3_canvas"]
    for t in range(3):
        out["c3_pan%d" % t] = (np.ascontiguousarray(cv[t:t + 376, 2 * t:2 * t + 1241]), 2000)
    out["c3_mosaic"] = (z["c3_mosaic"], 2000)
    out["c3_hubble"] = (z["c3_hubble"], 2000)
    out["stereo_l"] = (z["stereo_l"], 1200)
    out["stereo_r"] = (z["stereo_r"], 1200)
    return out


def oracle_digests(frames):
    from oracle import binding as ob
    ob.build()
    d = {}
    for name, (img, nf) in frames.items():
        r = ob.Extractor(nf, 1.2, 8, 20, 7)(img)
        d[name] = "%d:%s:%s" % (len(r["kps"]), digest(r["kps"]), digest(r["desc"]))
    return d


def main():
    warnings.filterwarnings("ignore")
    import skimage.data as sd
    z = {}
    for name in ("camera", "astronaut", "brick", "grass", "coffee", "rocket"):
        z["c2_" + name] = resized(getattr(sd, name)(), 480, 640)
    ret = grey(sd.retina())
    z["c3_canvas"] = np.ascontiguousarray(ret[520:520 + 392, 50:50 + 1305])
    cam, ast, gra = grey(sd.camera()), grey(sd.astronaut()), grey(sd.gravel())
    z["c3_mosaic"] = np.ascontiguousarray(np.concatenate([cam[60:436, 40:454], ast[20:396, 60:474], gra[100:476, 50:463]], axis=1))
    hub = sd.hubble_deep_field()
    z["c3_hubble"] = np.ascontiguousarray(resized(hub, 1082, 1241)[300:676])
    left, right, _ = sd.stereo_motorcycle()
    z["stereo_l"], z["stereo_r"] = grey(left), grey(right)
    for k, v in z.items():
        assert v.dtype == np.uint8 and v.flags["C_CONTIGUOUS"], k
    assert z["c3_mosaic"].shape == (376, 1241) and z["c3_hubble"].shape == (376, 1241) and z["stereo_l"].shape == (500, 741)
    dig = oracle_digests(frames_of(z))
    names = sorted(dig)
    z["digest_names"] = np.array(names)
    z["digest_values"] = np.array([dig[n] for n in names])
    out = os.path.join(ROOT, "tests", "golden", "natural.npz")
    np.savez_compressed(out, **z)
    print("wrote %s: %.2f MB, %d frames" % (out, os.path.getsize(out) / 1e6, len(names)))
    for n in names:
        print("  %-12s %s" % (n, dig[n][:40]))


if __name__ == "__main__":
    main()
