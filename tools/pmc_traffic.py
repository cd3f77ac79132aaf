#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes
(separate --pmc runs, ORBX_SERIAL=1).  fetch_kb / write_kb are KB per 64-frame step as reported, summed over the
kernel's dispatches of a step (frames_per_launch = 64 names that unit); fetch_scale is the calibration of tools/ubench/fetch_calib.hip on this box (profiles/r01_fetch_calibration.txt):
FETCH_SIZE reports exactly half of a streaming read at 1, 4 and 16 bytes per lane alike, WRITE_SIZE is exact --
bench.py uses fetch_kb * fetch_scale + write_kb."""
import json
import sqlite3
import sys


def short(name):
    return name.split("(")[0].split("::")[-1].split("<")[0]  # "void orbx::k_fast<48>(...)" -> "k_fast"


def per_step(rows):
    """rows: (kernel, dispatches, average value per dispatch) -> KB per 64-frame STEP.  A kernel that is dispatched
    more than once per step (k_fast: level 0, then levels 1-7) must be summed over its dispatches: the per-dispatch
    average of such a kernel is a fraction of the step's traffic (it made bench.py's `traffic` of k_fast 2x too small).
    Steps profiled = dispatches of k_pyramid, which runs once per step in ORBX_SERIAL=1."""
    rows = [(short(k), n, v) for k, n, v in rows]
    nsteps = max(n for k, n, v in rows if k == "k_pyramid")   # one launch per step when ORBX_SERIAL=1 (the blur's name depends on its form)
    out = {}
    for k, n, v in rows:
        out[k] = out.get(k, 0.0) + v * n / nsteps
    return out


def from_db(db, counter):
    c = sqlite3.connect(db)
    return per_step(c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall())


def from_table(path):
    """the text table tools/rocprof_summary.py pmc wrote from such a database (kernel, counter, calls, avg_value, avg_dur_ns)"""
    rows = []
    for line in open(path):
        f = line.split()
        if len(f) >= 5 and f[-4] in ("FETCH_SIZE", "WRITE_SIZE"):
            rows.append((" ".join(f[:-4]), int(f[-3]), float(f[-2])))
    return per_step(rows)


load = lambda p, c: from_table(p) if p.endswith(".txt") else from_db(p, c)
fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
res = {k: {"fetch_kb": fetch[k], "fetch_scale": 2.0, "write_kb": write.get(k, 0.0), "frames_per_launch": 64}
       for k in fetch if not k.startswith("__amd")}
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernels_sha import kernels_sha  # noqa: E402
res["_kernels_sha16"] = kernels_sha()   # bench.py replays this table: it says so when the kernels have changed since
json.dump(res, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
