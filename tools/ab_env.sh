#!/bin/bash
# A/B of an environment switch of the library inside ONE gpurun call: bash tools/ab_env.sh VAR [rounds]
# prints value (frames/s) and the isolated per-step kernel times of the matcher and FAST with VAR unset (A) and VAR=1 (B)
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; N=${2:-3}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export $VAR=1; else unset $VAR; fi
    python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('$v', round(d['value']), round(d['ms_per_step'],4), {n: round(k[n],4) for n in ('k_match_mfma','k_fast','k_blur','k_pyramid','k_orient_desc','k_distribute') if n in k})"
  done
done
