#!/bin/bash
# on the GPU box: bash tools/down_engine_ab.sh [rounds] ["ENV=.. ENV=.." variants, '|'-separated]
# default: the pipelined host entries with the results written by the shader (ORBX_DOWN_ENGINE=0) and gathered in HBM + one
# engine copy (1), alternating in one call
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}
VARS=${2:-"ORBX_DOWN_ENGINE=0|ORBX_DOWN_ENGINE=1"}
IFS='|' read -ra V <<< "$VARS"
for i in $(seq $N); do
  for v in "${V[@]}"; do echo -n "[$v] "; env $v python $R/tools/down_engine_ab.py 2>&1 | tail -3; done
done
