#!/usr/bin/env python3
"""Is there a real OpenCV on this box?  If so, hold the oracle's four restated OpenCV primitives to it.

VERDICT r5 #1: the oracle restates cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 (ORBextractor.cc:1120,
:809/:814, :1086, :103) from OpenCV 3.0's published algorithm; nothing in the tree has run beside a real OpenCV.
This script (a) looks for one -- every python interpreter's `import cv2`, pkg-config, shared objects anywhere on
the filesystem, a pip index -- and prints what it finds, and (b) when `cv2` imports, runs the comparisons of
tools/check_vs_opencv/check_vs_opencv.cpp stages 1-4 through it on the repository's frames and prints a per-stage
equal / differ table with the build's version.  It never substitutes anything for OpenCV: no cv2, no comparison.

usage: opencv_probe.py [--out FILE]      (exit 0 always: the output is the result)
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_lines = []


def say(s=""):
    print(s)
    _lines.append(s)


def run(cmd, timeout=60):
    try:
        p = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        return p.returncode, p.stdout.decode(errors="replace").strip()
    except subprocess.TimeoutExpired:
        return 124, "(timed out after %d s)" % timeout


def probe():
    say("== probe for an OpenCV installation ==")
    rc, out = run("uname -a; cat /etc/os-release | head -2")
    say(out)
    interps = sorted(set(p for pat in ("/usr/bin/python*", "/usr/local/bin/python*", "/opt/*/bin/python*", "/opt/conda/bin/python*",
                                       "/root/*/bin/python*") for p in glob.glob(pat) if os.access(p, os.X_OK) and "config" not in p))
    say("interpreters: %s" % " ".join(interps))
    found = False
    for py in interps:
        rc, out = run("%s -c \"import cv2; print(cv2.__version__)\"" % py, 90)
        last = out.splitlines()[-1] if out else ""
        say("  %s: import cv2 -> rc %d: %s" % (py, rc, last))
        found = found or rc == 0
    for pc in ("opencv4", "opencv"):
        rc, out = run("pkg-config --modversion %s" % pc)
        say("  pkg-config --modversion %s -> rc %d: %s" % (pc, rc, out.splitlines()[-1] if out else ""))
    rc, out = run("find / -xdev \\( -name 'libopencv_*' -o -name 'cv2*.so' -o -name 'cv2' -o -name 'opencv*.pc' -o -name 'opencv2' \\) "
                  "-not -path '*/mock_opencv/*' -not -path '/proc/*' 2>/dev/null | head -40", 240)
    say("  find / (libopencv_*, cv2*.so, cv2/, opencv*.pc, opencv2/ outside tests/cpp/mock_opencv): %s" % (out if out else "nothing"))
    rc, out = run("ldconfig -p | grep -i opencv")
    say("  ldconfig -p | grep opencv: %s" % (out if out else "nothing"))
    rc, out = run("%s -m pip download --no-deps -d /tmp/_cvprobe opencv-python-headless 2>&1 | tail -2" % sys.executable, 60)
    say("  pip download opencv-python-headless -> %s" % out.replace("\n", " | "))
    rc, out = run("%s -m pip list 2>/dev/null | grep -i -E 'opencv|scikit-image|kornia|imageio|pillow'" % sys.executable)
    say("  pip list (image libraries): %s" % (out.replace("\n", ", ") if out else "none"))
    shutil.rmtree("/tmp/_cvprobe", ignore_errors=True)
    return found


def compare():
    import numpy as np
    import cv2
    from oracle import binding as ob
    from orbslamm_amd import synth
    say("")
    say("== comparison: oracle (restated OpenCV 3.0 generic path) vs cv2 %s ==" % cv2.__version__)
    info = cv2.getBuildInformation()
    for ln in info.splitlines():
        if any(k in ln for k in ("Version control", "CPU/HW features", "Baseline", "Dispatched", "IPP", "OpenCL", "Built as dynamic")):
            say("  build: " + ln.strip())
    cv2.setUseOptimized(True)
    frames = {}
    for w, h in ((640, 480), (1241, 376), (401, 263)):
        for t, img in enumerate(synth.make_frames(w, h, 2)):
            frames["synth_%dx%d_%d" % (w, h, t)] = img
    try:
        from natural_cases import load
        for name, (img, _) in load()[0].items():
            frames["natural_" + name] = img
    except Exception as e:  # the natural fixture is optional for this tool
        say("  (natural frames not loaded: %s)" % e)

    tally = {k: [0, 0, 0] for k in ("resize", "fast20", "fast7", "blur", "atan2")}   # compared, differing, max |diff|

    # stage 4: fastAtan2 on check_vs_opencv.cpp's grid (moments of IC_Angle are integers up to 15*255*749)
    ys, xs = np.meshgrid(np.arange(-2000, 2001, 7, dtype=np.float32), np.arange(-2000, 2001, 5, dtype=np.float32), indexing="ij")
    for s in (1.0, 1431.0):
        y = (ys * np.float32(s)).ravel()
        x = (xs * np.float32(s)).ravel()
        a = cv2.phase(x, y, angleInDegrees=True).ravel().astype(np.float32)     # the array form of fastAtan2 (same kernel: FastAtan2_32f)
        b = np.array([cv2.fastAtan2(float(yy), float(xx)) for yy, xx in zip(y[::37], x[::37])], dtype=np.float32)
        L = ob.lib()
        import ctypes as C
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        o_all = np.array([L.orc_fast_atan2(float(yy), float(xx)) for yy, xx in zip(y[::37], x[::37])], dtype=np.float32)
        d = (b.view(np.uint32) != o_all.view(np.uint32))
        tally["atan2"][0] += d.size
        tally["atan2"][1] += int(d.sum())
        if d.any():
            tally["atan2"][2] = max(tally["atan2"][2], float(np.abs(b - o_all).max()))
            i = int(np.argmax(d))
            say("  fastAtan2 differs at (y=%g, x=%g): cv2 %.9g oracle %.9g  [cv2.phase there: %.9g]" % (y[::37][i], x[::37][i], b[i], o_all[i], a[::37][i]))

    for name, img in frames.items():
        H, W = img.shape
        nf = 2000 if W >= 1000 else 1000
        ex = ob.Extractor(nf, 1.2, 8, 20, 7)
        prev = np.ascontiguousarray(img)
        for l in range(8):
            lw, lh = ex.level_size(W, H, l)
            if l == 0:
                lvl = prev
            else:
                lvl = cv2.resize(prev, (lw, lh), interpolation=cv2.INTER_LINEAR)
                o = ob.resize(prev, lw, lh)
                d = lvl.astype(np.int16) - o.astype(np.int16)
                nb = int((d != 0).sum())
                tally["resize"][0] += d.size; tally["resize"][1] += nb; tally["resize"][2] = max(tally["resize"][2], int(np.abs(d).max()))
                if nb:
                    yy, xx = np.argwhere(d != 0)[0]
                    say("  resize %s level %d: %d of %d differ, first (%d,%d) cv2 %d oracle %d" % (name, l, nb, d.size, xx, yy, lvl[yy, xx], o[yy, xx]))
            for thr, key in ((20, "fast20"), (7, "fast7")):
                det = cv2.FastFeatureDetector_create(threshold=thr, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
                kp = det.detect(lvl, None)
                a = sorted((int(round(k.pt[1])), int(round(k.pt[0])), int(round(k.response))) for k in kp)
                oc = ob.fast(lvl, thr)
                b = sorted((int(c["y"]), int(c["x"]), int(c["score"])) for c in oc)
                sa, sb = set(a), set(b)
                nb = len(sa ^ sb)
                tally[key][0] += max(len(a), len(b)); tally[key][1] += nb
                if nb:
                    pa, pb = set(t[:2] for t in a), set(t[:2] for t in b)
                    say("  FAST t=%d %s level %d: cv2 %d corners, oracle %d; positions only-cv2 %d only-oracle %d; same position other score %d; e.g. %s" %
                        (thr, name, l, len(a), len(b), len(pa - pb), len(pb - pa), len((sa ^ sb)) - len(pa ^ pb) * 1, sorted(sa ^ sb)[:3]))
            blur = cv2.GaussianBlur(lvl, (7, 7), 2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
            o = ob.gaussian7(lvl)
            d = blur.astype(np.int16) - o.astype(np.int16)
            nb = int((d != 0).sum())
            tally["blur"][0] += d.size; tally["blur"][1] += nb; tally["blur"][2] = max(tally["blur"][2], int(np.abs(d).max()))
            prev = lvl
    say("")
    say("| stage (reference call site) | compared | differing | max abs diff | verdict |")
    say("|---|---|---|---|---|")
    names = {"resize": "cv::resize INTER_LINEAR (ORBextractor.cc:1120)", "fast20": "cv::FAST 9/16 t=20 + NMS (:809)", "fast7": "cv::FAST 9/16 t=7 + NMS (:814)",
             "blur": "cv::GaussianBlur 7x7 s=2 REFLECT_101 (:1086)", "atan2": "cv::fastAtan2 (:103)"}
    for k in ("resize", "fast20", "fast7", "blur", "atan2"):
        c, d, m = tally[k]
        say("| %s | %d | %d | %s | %s |" % (names[k], c, d, m, "EQUAL" if d == 0 else "DIFFERS"))
    say("frames: %d (%s ...)" % (len(frames), ", ".join(list(frames)[:4])))


def main():
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    found = probe()
    if found:
        try:
            compare()
        except Exception as e:
            import traceback
            say("comparison failed: %s" % e)
            say(traceback.format_exc())
    else:
        say("")
        say("RESULT: no OpenCV of any version on this box (no cv2 under any interpreter, no libopencv_*, no pkg-config entry, no pip index).")
        say("        The four restated OpenCV primitives stay unpinned; nothing was substituted for OpenCV.")
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "w") as f:
            f.write("\n".join(_lines) + "\n")


if __name__ == "__main__":
    main()
