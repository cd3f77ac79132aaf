#!/bin/bash
# HBM traffic of the stream matcher per 64-pair step (FETCH_SIZE / WRITE_SIZE passes, kernels alone): bash tools/match_traffic.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
pmc() { ORBX_SERIAL=1 timeout 240 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py --no-cpu-baseline --no-profile --no-host-path --no-tracking-path --no-parity-check --no-live-streams --pool 2 --steps 5 --warmup 2 > /dev/null 2>&1; }
pmc f FETCH_SIZE; pmc w WRITE_SIZE
python $R/tools/pmc_traffic.py /tmp/pmc_f/p_results.db /tmp/pmc_w/p_results.db /tmp/traffic.json > /dev/null
python - <<PY
import json
d=json.load(open('/tmp/traffic.json'))
for k in ('k_match_mfma','k_match_prune','k_orient_desc','k_roll_prev'):
    e=d[k]; print(k, 'fetch x2 + write = %.1f MB per step' % ((e['fetch_kb']*2+e['write_kb'])/1e3), '(fetch %.1f MB, write %.1f MB)' % (e['fetch_kb']*2/1e3, e['write_kb']/1e3))
PY
