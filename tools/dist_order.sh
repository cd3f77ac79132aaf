# on the GPU box: world 1 through the N > 1 path of bench.py (ORBX_BENCH_FORCE_DIST=1: torch + a 1-rank RCCL group)
# against the plain run: same throughput, and exactly one line on stdout, or the driver's N = 2/4/8 runs are off.
# History: with torch + RCCL initialised BEFORE the extractor handle a rank ran at 105 k instead of 129 k frames/s
# (the handle's four streams want the process's first four hardware queues; orbslamm_hip.hip orbx_create)
R=${GRAFT_REPO_ROOT:-/root/repo}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2 3; do
for v in plain dist; do
  if [ $v = dist ]; then export ORBX_BENCH_FORCE_DIST=1; else unset ORBX_BENCH_FORCE_DIST; fi
  timeout 300 python $R/bench.py --no-cpu-baseline --no-replay 2>$R/gpurun_out/dist_order_$v.err > $R/gpurun_out/dist_order_$v.out
  echo "$v stdout lines: $(wc -l < $R/gpurun_out/dist_order_$v.out)"
  tail -1 $R/gpurun_out/dist_order_$v.out | python -c "
import json,sys
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$v', round(d['value']), round(d['ms_per_step'],4), d['matches_last_frame'])
except Exception as e:
    print('$v', 'NO JSON', repr(t[:200]))"
done; done
tail -5 $R/gpurun_out/dist_order_dist.err
