#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-8s thr %2d x %d d%d att%d %8.0f frames/s  median %.3f mean %.3f p99 %.3f' % (d['mode'], d['robots'], d['cameras_per_call'], d['depth'], d['attach'], d['frames_per_s'], d['ms_median'], d['ms_mean'], d['ms_p99']))"; }
for pr in 1 0; do
  export ORBX_LAT_PRIO=$pr; echo "== ORBX_LAT_PRIO=$pr"
  $X --mode track --interval 0 --json | J
  $X --mode track --depth 2 --attach 0 --interval 0 --json | J
  $X --mode track --depth 1 --attach 0 --interval 0 --json | J
  for k in 4 6 8; do $X --mode track --robots $k --frames 400 --interval 0 --json | J; done
  $X --mode track --robots 4 --per-call 2 --frames 400 --interval 0 --json | J
  $X --mode bf --interval 0 --json | J
done
