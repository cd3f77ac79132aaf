#!/bin/bash
# on the GPU box: the native live-stream loops (examples/multi_robot) -- one robot, then K robots on the one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
for ls in ${LAT_LIST:-1 2}; do
  export ORBX_LAT_STREAMS=$ls
  echo "== ORBX_LAT_STREAMS=$ls GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-default}"
  $X --mode track --robots 1 --json
  $X --mode track --robots 1 --pinned 0 --json
  $X --mode track --robots 1 --depth 2 --json
  $X --mode track --robots 1 --depth 2 --attach 0 --json
  $X --mode bf --robots 1 --json
  $X --mode bf --robots 1 --pinned 0 --json
  $X --mode extract --robots 1 --json
  for k in 2 4 8 16; do $X --mode track --robots $k --frames 400 --json; done
  for k in 4 8; do $X --mode bf --robots $k --frames 400 --json; done
done
