#!/usr/bin/env python3
"""per-kernel table of PMC counters (average per dispatch) from a rocprofv3 rocpd db"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name").fetchall()
tab = defaultdict(dict)
ctrs = []
for k, cn, n, v, d in rows:
    k = k.split("(")[0]
    tab[k][cn] = v
    tab[k]["_calls"] = n
    tab[k]["_dur_us"] = d / 1e3
    if cn not in ctrs:
        ctrs.append(cn)
print("%-26s %6s %9s " % ("kernel", "calls", "dur_us") + " ".join("%16s" % x[-16:] for x in ctrs))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["_dur_us"]):
    if k.startswith("__amd"):
        continue
    print("%-26s %6d %9.1f " % (k[-26:], v["_calls"], v["_dur_us"]) + " ".join("%16.1f" % v.get(x, float("nan")) for x in ctrs))
