#!/bin/bash
# on the GPU box: kernel timeline of two consecutive one-frame-per-call invocations (extract + match), from rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/b1prof -o t -- python $R/tools/b1_loop.py 40 > /dev/null 2>&1
python $R/tools/timeline.py /tmp/b1prof/t_results.db 28 14
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/b1prof/t_results.db")
try:
    rows = c.execute("select start, end, name from memory_copies order by start").fetchall()[-8:]
    t0 = rows[0][0]
    for s, e, n in rows:
        print("copy %9.1f %9.1f %7.1f %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
except Exception as e:
    print("no copy table:", e)
PY
