#!/bin/bash
# on the GPU box: what limits K robots on one GPU?  threads vs processes, hardware queues, launches per frame
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-8s robots %2d  %8.0f frames/s  median %.3f mean %.3f p99 %.3f  host submit %.0f enqueue %.0f us' % (d['mode'], d['robots'], d['frames_per_s'], d['ms_median'], d['ms_mean'], d['ms_p99'], d['host_us_submit'], d['host_us_enqueue']))"; }
for ls in 1 2; do
export ORBX_LAT_STREAMS=$ls
for q in 4 8 16; do
  export GPU_MAX_HW_QUEUES=$q
  echo "== threads, track, ORBX_LAT_STREAMS=$ls GPU_MAX_HW_QUEUES=$q"; for k in 1 2 4 8 16; do $X --mode track --robots $k --frames 400 --interval 0 --json | J; done
done
done
export ORBX_LAT_STREAMS=1 GPU_MAX_HW_QUEUES=8
echo "== threads, extract / bf, 1 stream, 8 queues"; for k in 4 8; do $X --mode extract --robots $k --frames 400 --interval 0 --json | J;  $X --mode bf --robots $k --frames 400 --interval 0 --json | J; done
unset GPU_MAX_HW_QUEUES
echo "== processes x 1 robot, track"
for k in 2 4 8; do
  for i in $(seq 1 $k); do $X --mode track --robots 1 --frames 3000 --warmup 300 --interval 0 --json > /tmp/p_$i.txt 2>&1 & done; wait
  cat /tmp/p_*.txt | grep '^{' | python -c "
import sys, json
d = [json.loads(l) for l in sys.stdin]
print('%d processes: sum %.0f frames/s, per process median %.3f ms mean %.3f' % (len(d), sum(x['frames_per_s'] for x in d), sum(x['ms_median'] for x in d) / len(d), sum(x['ms_mean'] for x in d) / len(d)))"
  rm -f /tmp/p_*.txt
done
