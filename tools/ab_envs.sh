#!/bin/bash
# A/B of several environment settings inside ONE gpurun call: bash tools/ab_envs.sh rounds "VAR=1" "VAR2=1 VAR3=1" ...
# ("-" = nothing set).  Prints frames/s over the default 200 steps, ms per step and the quadtree's / FAST's isolated times.
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then E=""; else E="$cfg"; fi
    env $E python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('[$cfg]', round(d['value']), round(d['ms_per_step'],4), 'parity', d.get('parity_check',{}).get('ok'), {n: round(k[n],4) for n in ('k_fast','k_distribute') if n in k})"
  done
done
