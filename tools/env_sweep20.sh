#!/bin/bash
# env_sweep.sh on the driver's invocation (--steps 20 --warmup 5): bash tools/env_sweep20.sh VAR rounds val1 val2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; N=$2; shift 2
for i in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', round(d['value']), round(d['ms_per_step'],4))"
  done
done
