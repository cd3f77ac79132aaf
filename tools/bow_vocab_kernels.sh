#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in complete orbvoc strong; do
  rm -rf /tmp/bowprof
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/bowprof -o t -- python $R/tools/bow_vocab_kernels.py $v 2>/dev/null | grep nodes
  python $R/tools/rocprof_summary.py stats /tmp/bowprof/t_results.db | grep -E "voc|bow"
done
