#!/usr/bin/env python3
"""Which kernels of the pipeline slow each other down when they share the GPU ("the contended
resource").  rocprofv3's counter mode serialises dispatches, so this is measured directly: kernel i back to back on one
stream while kernel j keeps another stream busy (orbx_debug_pair_overlap), at the sub-batch size of the benchmark
(32 frames per launch).  Next to the slowdowns: each kernel's residency limits per CU from its footprint
(VGPRs from the code object, LDS per workgroup from the launch), i.e. what a co-runner takes away.
usage (GPU box): python tools/pair_overlap.py [frames_per_launch] > profiles/rNN_pair_overlap.txt"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslamm_amd import ORBextractor, synth  # noqa: E402
from orbslamm_amd._lib import check  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W, H, B = 1241, 376, 64
frames = synth.make_frames(W, H, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B)
d = ex.upload_frames(frames, stride=1280)
for _ in range(3):
    ex.extract_batch_device(*d)
    ex.match_prev_batch_device(0.7, 50, True)
ex.sync()
K = 6
n = C.c_int(0)
names = (C.c_char_p * K)()
alone = np.zeros(K, np.float32)
co = np.zeros((K, K), np.float32)
lds = np.zeros(K, np.int32); thr = np.zeros(K, np.int32); wgs = np.zeros(K, np.int32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
reps = []
for _ in range(3):
    check(ex._L.orbx_debug_pair_overlap(ex._h, NB, C.c_float(1.5), C.byref(n), names, P(alone), P(co), P(lds), P(thr), P(wgs)))
    reps.append((alone.copy(), co.copy()))
alone = np.median([r[0] for r in reps], axis=0)
co = np.median([r[1] for r in reps], axis=0)
nm = [names[i].decode() for i in range(K)]

# residency limits per CU: 4 SIMDs x 512 VGPRs per lane, 8 waves per SIMD at most, 160 KB of LDS
vg = {}
try:
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], stderr=subprocess.DEVNULL).decode()
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+)", line)
        if m:
            for k in nm:
                key = {"k_match_mfma": "12k_match_mfma", "k_fast": "k_fastILi48", "k_distribute": "k_distributeILb1"}.get(k, k)
                if key in m.group(1):
                    vg[k] = int(m.group(2))
except Exception:
    pass
print("# co-run slowdowns at %d frames per launch (MI355X, median of 3): time per launch of ROW beside COLUMN / alone" % NB)
print("%-15s %9s  " % ("kernel", "alone ms") + " ".join("%13s" % x[2:] for x in nm))
for i in range(K):
    print("%-15s %9.4f  " % (nm[i], alone[i]) + " ".join("%13.2f" % (co[i, j] / alone[i]) for j in range(K)))
print()
print("# footprint: workgroups per launch, threads and LDS per workgroup, VGPRs per lane -> resident waves per CU allowed by")
print("# LDS (160 KB / LDS per workgroup x waves per workgroup) and by VGPRs (4 SIMDs x min(8, 512 // VGPRs)); a CU holds 32 waves")
print("%-15s %8s %8s %9s %6s %10s %10s" % ("kernel", "wgs", "threads", "lds B", "vgpr", "waves(LDS)", "waves(VGPR)"))
for i in range(K):
    wpw = thr[i] // 64
    by_lds = min(32, (160 * 1024 // max(int(lds[i]), 1)) * wpw)
    v = vg.get(nm[i], 0)
    by_v = 4 * min(8, 512 // max((v + 7) // 8 * 8, 8)) if v else -1
    print("%-15s %8d %8d %9d %6d %10d %10d" % (nm[i], wgs[i], thr[i], lds[i], v, by_lds, by_v))
print()
s = co / alone[:, None]
print("# pair cost = (slowdown of A beside B) x (slowdown of B beside A): 1 = they overlap for free, 4 = they gain nothing over running one after the other")
for i in range(K):
    for j in range(i + 1, K):
        tot = s[i, j] * s[j, i]
        print("%-15s + %-15s  %5.2f x %5.2f = %5.2f" % (nm[i], nm[j], s[i, j], s[j, i], tot))
