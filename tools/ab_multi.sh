#!/bin/bash
# A/B/C... throughput of several builds of the library inside ONE gpurun call: A = the in-tree .so, every other letter
# build_ub/lib<letter>.so.  usage (gpurun): bash tools/ab_multi.sh "A B C" [rounds] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-"A B"}; N=${2:-3}; shift 2
for i in $(seq $N); do
  for v in $V; do
    if [ $v = A ]; then unset ORBSLAMM_HIP_LIB; else export ORBSLAMM_HIP_LIB=$R/build_ub/lib$v.so; fi
    python $R/bench.py --no-cpu-baseline --no-replay --no-host-path "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done
