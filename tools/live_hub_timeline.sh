#!/bin/bash
# on the GPU box: kernel timeline of the chain with P cameras per call (examples/multi_robot --per-call P, one thread)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
cd /tmp && export TMPDIR=/tmp
export ORBX_LAT_MAXB=8
for p in ${PLIST:-1 4 8}; do
  rm -rf /tmp/hubprof
  timeout 200 rocprofv3 --kernel-trace -d /tmp/hubprof -o t -- $X --mode track --robots 1 --per-call $p --frames 60 --warmup 20 --interval 0 --json > /dev/null 2>&1
  echo "== per-call $p"
  python3 $R/tools/timeline.py /tmp/hubprof/t_results.db 12 24
done
