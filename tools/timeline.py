#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 --kernel-trace rocpd database:
start/end (us, relative to the step's first kernel), duration, kernel, queue.  usage: timeline.py <db> [nkernels [skip_last]]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
if not cols:
    print([r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")])
    sys.exit(1)
rows = c.execute("select start, end, name, queue_id, stream_id from kernels order by start").fetchall() if "stream_id" in cols else \
       [r + (0,) for r in c.execute("select start, end, name, queue_id from kernels order by start").fetchall()]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # records to drop from the end (skip the drain of the last steps)
rows = rows[-(n + skip):len(rows) - skip] if skip else rows[-n:]
t0 = rows[0][0]
for s, e, name, q, st in rows:
    print("%9.1f %9.1f %8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, name.split("(")[0][:40]))
