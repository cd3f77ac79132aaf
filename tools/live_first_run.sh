#!/bin/bash
# on the GPU box: does the very first live-stream run of a fresh box behave (bench.py's live_streams block timed out on it once)?
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
for i in 1 2 3; do
  SECONDS=0
  timeout 120 $X --json --interval 0 --w 1241 --h 376 --nfeat 2000 --gpus 1 --frames 600 --warmup 40 --mode track 2>&1 | tail -2 | cut -c1-300
  echo "took $SECONDS s"
done
for i in 1 2; do
  SECONDS=0
  timeout 600 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-tracking-path --no-parity-check --no-profile > /tmp/b$i.json 2>/tmp/b$i.err; tail -1 /tmp/b$i.err
  python3 - <<PY
import json
d=json.loads(open("/tmp/b$i.json").read().strip().splitlines()[-1])
print({k:(v.get("ms_median"), v.get("error","")[:80]) for k,v in d["live_streams"]["one_robot"].items()})
PY
  echo "took $SECONDS s"
done
