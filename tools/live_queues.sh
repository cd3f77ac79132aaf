#!/bin/bash
# on the GPU box: which hardware queue does each robot's chain run on?  (rocprofv3 kernel trace of K robots)
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qprof
timeout 200 rocprofv3 --kernel-trace -d /tmp/qprof -o t -- $R/examples/multi_robot --mode track --robots $K --frames 60 --warmup 10 --interval 0 --json 2>/dev/null | grep '^{' | cut -c1-200
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/qprof/t_results.db")
qs = {}
for q, st, n in c.execute("select queue_id, stream_id, count(*) from kernels where name like '%k_pyramid%' group by queue_id, stream_id"):
    qs.setdefault(q, []).append((st, n))
print("k_pyramid launches per (queue: [(stream, count)]):", qs)
PY
