#!/bin/bash
# on the GPU box: the live-stream loop (tools/live_loop.py) timed, then its kernel timeline under rocprofv3
# usage: live_timeline.sh [mode=track] [nkernels=30]
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-track}
cd /tmp && export TMPDIR=/tmp
python $R/tools/live_loop.py $MODE 600
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/liveprof -o t -- python $R/tools/live_loop.py $MODE 40 > /dev/null 2>&1
python $R/tools/timeline.py /tmp/liveprof/t_results.db ${2:-30} 14
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/liveprof/t_results.db")
try:
    rows = c.execute("select start, end, name from memory_copies order by start").fetchall()[-8:]
    t0 = rows[0][0]
    for s, e, n in rows:
        print("copy %9.1f %9.1f %7.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
except Exception as e:
    print("no copy table:", e)
PY
