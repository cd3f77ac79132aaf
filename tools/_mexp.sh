R=${GRAFT_REPO_ROOT:-/root/repo}
for v in old new old new old new; do
  if [ $v = new ]; then unset ORBSLAMM_HIP_LIB; else export ORBSLAMM_HIP_LIB=$R/build_ub/libA.so; fi
  timeout 120 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v 20 steps', round(d['value']), round(d['ms_per_step'],4))"
done
