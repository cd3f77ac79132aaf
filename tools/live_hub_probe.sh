#!/bin/bash
# on the GPU box: how long is the latency-mode chain with P cameras per call (ORBX_LAT_MAXB=8), alone and with several threads;
# frames of different cameras go up by one copy kernel (default) or one engine copy each (ORBX_LAT_DMA=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
export ORBX_LAT_MAXB=8
run() { $X --mode track --robots $2 --per-call $1 --frames 400 --json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per-call $1 threads $2 dma=${ORBX_LAT_DMA:-0}', {k:d[k] for k in d if k in ('frames_per_s','ms_median','ms_p99','host_us_submit')})"; }
for rep in 1 2; do
for p in ${PLIST:-2 4 8}; do for k in ${KLIST:-1 2 4}; do
  ORBX_LAT_DMA=0 run $p $k; ORBX_LAT_DMA=1 run $p $k
done; done; done
