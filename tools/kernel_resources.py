#!/usr/bin/env python3
"""VGPR / SGPR / LDS / spill figures of every kernel of the library (cross-compiles, no GPU needed):
hipcc --offload-device-only -S, then the amdhsa metadata."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:]
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-device-only", "-S",
                       "-o", "/tmp/orbx_k.s", os.path.join(root, "orbslamm_amd", "csrc", "orbslamm_hip.hip")] + extra,
                      stderr=subprocess.DEVNULL)
txt = open("/tmp/orbx_k.s").read()
md = txt[txt.index("amdhsa.kernels"):]
for k in md.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", k).group(1)
    g = lambda f: re.search(r"\.%s:\s+(\d+)" % f, k).group(1)
    print("%-64s vgpr %3s sgpr %3s lds %6s spills v%s s%s scratch %s" % (name[:64], g("vgpr_count"), g("sgpr_count"),
          g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
