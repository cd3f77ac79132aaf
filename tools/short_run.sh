# how the timed value depends on the length of the run and on what ran before it (one gpurun call): bash tools/short_run.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-replay "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],4))"; }
for i in 1 2; do
run --steps 20 --warmup 5
run --steps 20 --warmup 50
run --steps 20 --warmup 200
run --steps 20 --warmup 1000
run --steps 40 --warmup 5
run --steps 200 --warmup 5
run --steps 1000 --warmup 5
done
