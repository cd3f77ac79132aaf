#!/bin/bash
# on the GPU box: bash tools/down_engine_ab.sh [rounds] -- the pipelined host entries with the results written by the shader
# (ORBX_DOWN_ENGINE=0) and gathered in HBM + one engine copy (1), alternating in one call
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}
for i in $(seq $N); do
  for v in 0 1; do ORBX_DOWN_ENGINE=$v python $R/tools/down_engine_ab.py 2>&1 | tail -1; done
done
