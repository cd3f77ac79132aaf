#!/bin/bash
# on the GPU box: R robots (one thread each, one frame per call) behind hubs of P cameras (examples/multi_robot --hub P) beside
# the same robots with a handle each
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', {k:d[k] for k in d if k in ('frames_per_s','ms_median','ms_p99','hub_batch_mean','matches_mean','checksum')})"; }
for r in ${RLIST:-8 16}; do
  $X --mode track --robots $r --frames 400 --json | J "robots $r, a handle each"
  for p in ${PLIST:-2 4 8}; do $X --mode track --robots $r --hub $p --frames 400 --json | J "robots $r, hubs of $p"; done
done
$X --mode track --robots 4 --hub 4 --frames 400 --json | J "robots 4, one hub"
$X --mode track --robots 2 --hub 2 --frames 400 --json | J "robots 2, one hub"
$X --mode track --robots 8 --hub 4 --pinned 0 --frames 400 --json | J "robots 8, hubs of 4, pageable frames"
