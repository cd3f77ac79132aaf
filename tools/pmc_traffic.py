#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes
(separate --pmc runs, ORBX_SERIAL=1, 64 frames per launch).  fetch_kb / write_kb are KB per launch as
reported; fetch_scale is the calibration of tools/ubench/fetch_calib.hip on this box (profiles/r01_fetch_calibration.txt):
FETCH_SIZE reports exactly half of a streaming read at 1, 4 and 16 bytes per lane alike, WRITE_SIZE is exact --
bench.py uses fetch_kb * fetch_scale + write_kb."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, v in c.execute("select kernel_name, avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        out[name.split("(")[0].split("::")[-1].split("<")[0]] = v  # "void orbx::k_fast<48>(...)" -> "k_fast"
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {k: {"fetch_kb": fetch[k], "fetch_scale": 2.0, "write_kb": write.get(k, 0.0), "frames_per_launch": 64}
       for k in fetch if not k.startswith("__amd")}
json.dump(res, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
