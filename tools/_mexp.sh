R=${GRAFT_REPO_ROOT:-/root/repo}
ORBSLAMM_HIP_LIB=$R/build_ub/libA.so timeout 600 python -m pytest $R/tests/test_gpu_matcher.py -x -q -m gpu 2>&1 | tail -1
for v in A B0 A B0 A B0; do
  export ORBSLAMM_HIP_LIB=$R/build_ub/lib$v.so
  timeout 120 python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']
print('$v', round(d['value']), round(d['ms_per_step'],4), round(k['k_match_mfma'],4))"
done
