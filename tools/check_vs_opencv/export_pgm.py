#!/usr/bin/env python3
"""Write the repository's test frames as binary PGM files for check_vs_opencv: the synthetic scenes of
orbslamm_amd/synth.py at the two benchmark shapes and the natural images of tests/golden/natural.npz.
usage: export_pgm.py <directory>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (w, h))
        f.write(img.tobytes())


def main():
    from orbslamm_amd import synth
    from natural_cases import load
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    for w, h in ((640, 480), (1241, 376), (401, 263)):
        for t, img in enumerate(synth.make_frames(w, h, 2)):
            write(os.path.join(out, "synth_%dx%d_%d.pgm" % (w, h, t)), img)
    for name, (img, _) in load()[0].items():
        write(os.path.join(out, "natural_%s.pgm" % name), img)
    print("wrote %d frames to %s" % (len(os.listdir(out)), out))


if __name__ == "__main__":
    main()
