#!/bin/bash
# sweep one environment switch of the library inside ONE gpurun call: bash tools/env_sweep.sh VAR rounds val1 val2 ...
# ("-" = unset); prints value (frames/s), ms per step, parity and the isolated per-step kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; N=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('$VAR=$v', round(d['value']), round(d['ms_per_step'],4), d.get('parity_check',{}).get('ok'), {n: round(k[n],4) for n in ('k_pyramid','k_fast','k_blur','k_orient_desc','k_distribute','k_match_mfma') if n in k})"
  done
done
