#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE rocprofv3 passes
(separate --pmc runs, ORBX_SERIAL=1, 64 frames per launch).  Values are KB per launch as
reported (FETCH_SIZE on gfx950 halves wide 16 B/lane streams -- MI355X_MICROARCH.md HBM;
our kernels read 1-4 B/lane, so the raw figure is kept and the caveat stated)."""
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
res = {k: {"fetch_kb": fetch[k], "write_kb": write.get(k, 0.0), "frames_per_launch": 64}
       for k in fetch if not k.startswith("__amd")}
json.dump(res, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
