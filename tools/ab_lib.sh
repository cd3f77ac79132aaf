#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call: bash tools/ab_lib.sh path/to/libA.so [rounds]
# A = the given library (ORBSLAMM_HIP_LIB), B = the in-tree build; prints value (frames/s) and the isolated per-step kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBA=$1; N=${2:-3}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then export ORBSLAMM_HIP_LIB=$R/$LIBA; else unset ORBSLAMM_HIP_LIB; fi
    python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('$v', round(d['value']), round(d['ms_per_step'],4), d.get('parity_check',{}).get('ok'), {n: round(k[n],4) for n in ('k_match_mfma','k_fast','k_blur','k_pyramid','k_orient_desc','k_distribute') if n in k})"
  done
done
