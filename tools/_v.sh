#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python $R/bench.py --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['isolated']['kernel_ms_per_step']['k_distribute'],4))"; }
for i in 1 2; do
  run pad0
  ORBX_DIST_LDS_PAD=16384 run pad16k
  ORBX_DIST_LDS_PAD=32768 run pad32k
  ORBX_DIST_LDS_PAD=65536 run pad64k
done
