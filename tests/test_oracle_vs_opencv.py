"""Opt-in pin of the oracle's four restated OpenCV primitives against a REAL OpenCV.

Runs only where `import cv2` works; everywhere else it skips, loudly, with the reason.  Nothing here stands in for
OpenCV: without the library there is no comparison.  Rounds 1-6 probed the build container and the MI355X boxes
(tools/opencv_probe.py, profiles/r06_opencv_probe.txt): no OpenCV of any version, no package index.

What a run means, per stage (reference call sites in ORBextractor.cc):
  cv::resize INTER_LINEAR u8 (:1120)   -- the generic fixed-point path has not changed since 3.0: must be EQUAL
  cv::FAST 9/16 + NMS, t = 20 / 7 (:809, :814) -- unchanged since 3.0: must be EQUAL (positions and responses)
  cv::fastAtan2 (:103)                 -- the scalar polynomial of 3.0 .. 4.x: must be EQUAL
  cv::GaussianBlur 7x7 sigma 2 u8 (:1086) -- 3.4+ switched 8-bit blurs to a fixed-point kernel with other rounding:
                                          a difference against >= 3.4 is EXPECTED and reported, not failed;
                                          against 3.0 .. 3.3 it must be EQUAL
The test is in the CPU suite and in the GPU suite (the comparison needs no GPU; the GPU box is simply another place
where an OpenCV might exist)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cv2():
    try:
        import cv2
        return cv2
    except Exception as e:  # ImportError, or a broken binary wheel
        pytest.skip("NO REAL OPENCV HERE (import cv2: %s) -- the oracle's cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 "
                    "restatements stay UNPINNED; see profiles/r06_opencv_probe.txt" % e)


def _run_probe(tmp_path):
    out = tmp_path / "probe.txt"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "opencv_probe.py"), "--out", str(out)], check=True,
                   stdout=subprocess.DEVNULL, timeout=900)
    return out.read_text()


def _table(text):
    rows = {}
    for ln in text.splitlines():
        if ln.startswith("| cv::"):
            c = [x.strip() for x in ln.strip("|").split("|")]
            rows[c[0].split(" (")[0]] = (int(c[1]), int(c[2]), c[4])
    return rows


def _check(tmp_path):
    cv2 = _cv2()
    text = _run_probe(tmp_path)
    rows = _table(text)
    assert len(rows) == 5, "the probe printed no comparison table:\n" + text[-2000:]
    ver = tuple(int(x) for x in cv2.__version__.split(".")[:2])
    must = [k for k in rows if "GaussianBlur" not in k]
    if ver < (3, 4):
        must = list(rows)
    bad = {k: rows[k] for k in must if rows[k][2] != "EQUAL"}
    assert not bad, "oracle differs from OpenCV %s where it must not: %s\n%s" % (cv2.__version__, bad, text[-3000:])
    blur = [rows[k] for k in rows if "GaussianBlur" in k][0]
    if blur[2] != "EQUAL":
        print("GaussianBlur differs from OpenCV %s in %d of %d pixels (expected for >= 3.4: another fixed-point kernel)" % (cv2.__version__, blur[1], blur[0]))


def test_oracle_primitives_against_a_real_opencv(tmp_path):
    _check(tmp_path)


@pytest.mark.gpu
def test_oracle_primitives_against_a_real_opencv_on_the_gpu_box(tmp_path):
    _check(tmp_path)


def test_probe_records_are_committed():
    """The negative result of this round's search is a committed artefact, not a claim."""
    p = os.path.join(ROOT, "profiles", "r06_opencv_probe.txt")
    text = open(p).read()
    assert "probe for an OpenCV installation" in text
    assert ("RESULT: no OpenCV" in text) or ("| cv::resize" in text)
