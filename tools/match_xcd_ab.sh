#!/bin/bash
# the stream matcher's block -> frame-pair map: pairs dealt round-robin to the XCDs (in-tree build) against runs of consecutive
# pairs per XCD (build_ub/libB.so = -DORBM_XCD_RUN=1): HBM traffic of k_match_mfma per 64-frame step (separate FETCH_SIZE /
# WRITE_SIZE passes, ORBX_SERIAL=1) and the throughput A/B.  One gpurun call: bash tools/match_xcd_ab.sh [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
pmc() { # tag counter
  ORBX_SERIAL=1 timeout 240 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py --no-cpu-baseline --no-profile --no-host-path --no-tracking-path --no-parity-check --no-live-streams --pool 2 --steps 5 --warmup 2 > /dev/null 2>&1
}
for v in A B; do
  if [ $v = B ]; then export ORBSLAMM_HIP_LIB=$R/build_ub/libB.so; else unset ORBSLAMM_HIP_LIB; fi
  pmc f$v FETCH_SIZE; pmc w$v WRITE_SIZE
  python $R/tools/pmc_traffic.py /tmp/pmc_f$v/p_results.db /tmp/pmc_w$v/p_results.db /tmp/traffic_$v.json > /dev/null
  python - <<PY
import json
d=json.load(open('/tmp/traffic_$v.json'))
for k in ('k_match_mfma','k_orient_desc'):
    e=d[k]; print('$v', k, 'fetch_kb x2 + write_kb = %.1f MB per step' % ((e['fetch_kb']*2+e['write_kb'])/1e3), e)
PY
done
unset ORBSLAMM_HIP_LIB
cd $R
N=${1:-3}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export ORBSLAMM_HIP_LIB=$R/build_ub/libB.so; else unset ORBSLAMM_HIP_LIB; fi
    python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('$v', round(d['value']), round(d['ms_per_step'],4), d.get('parity_check',{}).get('ok'), {n: round(k[n],4) for n in ('k_match_mfma','k_fast','k_pyramid') if n in k})"
  done
done
