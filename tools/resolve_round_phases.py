import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from orbslamm_amd import ORBextractor, ORBmatcher, make_grid, synth, _lib
W, H, NF = 1241, 376, 2000
fr = synth.make_frames(W, H, 2, stream=0)
ex = ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2, device=0)
m = ORBmatcher(0.9, True, device=0)
sf = np.array(ex.GetScaleFactors(), np.float32)
g = make_grid(0.0, 0.0, float(W), float(H))
fs = m.frame_set(2, ex.max_keypoints, [718.856, 718.856, 607.1928, 185.2157], [0] * 5, g, [0.0, float(W), 0.0, float(H)], sf)
ex.extract_batch_device(*ex.upload_frames(fr))
fs.build_from_extractor(0, ex)
L = _lib.lib()
for th in (15.0, 30.0):
    acc = np.zeros(15)
    for rep in range(20):
        fs.track([1], [0], th=th); fs.results()
        us = (C.c_double * 15)(); L.orbm_debug_phase_times(us, 15)
        if rep >= 5: acc += np.array(list(us))
    acc /= 15
    print("th=%g round 1: post loop %.2f, barrier(+runner-up) %.2f, decide loop %.2f, clear+barrier %.2f us; init %.1f rounds %.1f writeback %.1f" % (th, acc[11], acc[12], acc[13], acc[14], acc[0], acc[1], acc[2]))
    cen = (C.c_int32 * 128)(); L.orbm_debug_round_census(cen, 64)
    rows = []
    for i in range(64):
        if cen[2 * i] < 0: break
        rows.append("%d/%d" % (cen[2 * i], cen[2 * i + 1]))
    print("   live entries / undecided queries per round:", " ".join(rows))
bu = (C.c_double * 11)(); L.orbm_debug_build_phases(bu, 8)
print("k_frame_build: clear %.2f, keys + undistort + count %.2f, descriptor copy %.2f, barrier %.2f, scan %.2f, scatter %.2f, sort %.2f, write %.2f us" % tuple(list(bu)[:8]))
