#!/bin/bash
# on the GPU box: result kernel inside the frame build's launch (default) or as a launch of its own (ORBX_FUSE_PACK=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', {k:d[k] for k in d if k in ('frames_per_s','ms_median','ms_p99','checksum')})"; }
for rep in 1 2 3; do for fp in 1 0; do
  export ORBX_FUSE_PACK=$fp
  $X --mode track --robots 1 --frames 1500 --interval 0 --json | J "fuse=$fp one robot       "
  $X --mode full --robots 1 --frames 1000 --interval 0 --json | J "fuse=$fp one robot, full "
  $X --mode track --robots 4 --hub 4 --frames 600 --interval 0 --json | J "fuse=$fp 4 robots, 1 hub "
  $X --mode track --robots 8 --hub 2 --frames 400 --interval 0 --json | J "fuse=$fp 8 robots, 4 hubs"
done; done
