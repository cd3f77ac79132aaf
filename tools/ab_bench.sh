#!/bin/bash
# A/B throughput of two builds of the library inside ONE gpurun call (box-to-box noise is ~1.5 %):
#   here:    hipcc ... -o build_ub/libB.so orbslamm_amd/csrc/orbslamm_hip.hip   (the variant; A = the in-tree .so)
#   gpurun:  bash tools/ab_bench.sh [rounds]
# e.g. the issue-sensitivity experiment of DESIGN.md section 5: B = the same source with -DORBX_EXP_INFLATE=100
# (k_blur executes 100 extra VALU instructions per thread, +10.7 M wave-instructions = +5 % of a step): -1.9 % over four rounds.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-4}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export ORBSLAMM_HIP_LIB=$R/build_ub/libB.so; else unset ORBSLAMM_HIP_LIB; fi
    python $R/bench.py --no-cpu-baseline --no-replay 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['value']), d['ms_per_step'])"
  done
done
