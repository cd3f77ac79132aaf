#!/bin/bash
# on the GPU box: the tracking-path figures that move with the candidate / resolve kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do $R/examples/multi_robot --mode track --interval 0 --json | grep "^{" | cut -c150-260; done
python $R/bench.py --no-cpu-baseline --no-host-path --no-live-streams --no-replay --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tracking_path']; print('b1', round(t['b1']['ms_median'],4), 'search64', round(t['batched']['search_only_ms_per_step'],4), 'pairs/s', round(t['batched']['pairs_per_s']), 'dropin', round(t['dropin']['ms_median'],4), 'local th1', round(t['local_map']['th1']['ms_median'],4), 'th3', round(t['local_map']['th3']['ms_median'],4), t['parity_ok'])"
