#!/bin/bash
# on the GPU box: K robots on one GPU under rocprofv3 --kernel-trace: do the robots' chains overlap on the device?
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-8}
cd /tmp && export TMPDIR=/tmp
for k in 1 4 $K; do $R/examples/multi_robot --mode track --robots $k --frames 400 --interval 0 --json | grep '^{'; done
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/k8prof -o t -- $R/examples/multi_robot --mode track --robots $K --frames 60 --warmup 10 --interval 0 --json > /dev/null 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/k8prof/t_results.db")
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select start, end, name, queue_id, stream_id from kernels order by start").fetchall()
n = len(rows)
rows = rows[n // 2: n // 2 + 110]
t0 = rows[0][0]
for s, e, name, q, st in rows:
    print("%9.1f %9.1f %8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, name.split("(")[0][:36]))
qs = {}
for s, e, name, q, st in c.execute("select start, end, name, queue_id, stream_id from kernels").fetchall():
    qs.setdefault(q, set()).add(st)
print({q: sorted(v) for q, v in qs.items()})
PY
