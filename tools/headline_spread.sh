# on the GPU box: the default bench command N times on one box -> spread of the headline figure
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-6}
for i in $(seq $N); do
  python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],4), round(r['avg_launch_ms'],4), round(r['isolated']['kernel_ms_per_step']['k_fast'],4))"
done | tee $R/gpurun_out/headline_spread.txt
python - <<PY
import statistics
v=[int(l.split()[0]) for l in open("$R/gpurun_out/headline_spread.txt")]
print("n", len(v), "min", min(v), "median", statistics.median(v), "max", max(v), "spread %.2f %%" % (100*(max(v)-min(v))/statistics.median(v)))
PY
