#!/bin/bash
# on the GPU box: one time base for the upload DMA and the chain's kernels of the native live loop (how long after the
# DMA's end does the first kernel start; how long after the host's submit does the DMA start)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dmaprof
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/dmaprof -o t -- $R/examples/multi_robot --mode ${1:-track} --frames 40 --warmup 10 --interval 0 --json > /dev/null 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/dmaprof/t_results.db")
ev = [(s, e, "COPY") for s, e in c.execute("select start, end from memory_copies")]
ev += [(s, e, n.split("(")[0].split("::")[-1][:22]) for s, e, n in c.execute("select start, end, name from kernels")]
ev.sort()
ev = ev[len(ev) * 2 // 3:]
i0 = next(i for i, x in enumerate(ev) if x[2] in ("COPY", "k_upload"))
t0 = ev[i0][0]
for s, e, n in ev[i0:i0 + 26]:
    print("%9.1f %9.1f %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
