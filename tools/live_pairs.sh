#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-8s threads %2d x %d cameras  %8.0f frames/s  median %.3f mean %.3f p99 %.3f  host submit %.0f enqueue %.0f us  kp %.0f matches %.0f' % (d['mode'], d['robots'], d['cameras_per_call'], d['frames_per_s'], d['ms_median'], d['ms_mean'], d['ms_p99'], d['host_us_submit'], d['host_us_enqueue'], d['keypoints_mean'], d['matches_mean']))"; }
for k in 1 2 3 4 6 8; do $X --mode track --robots $k --frames 400 --interval 0 --json | J; done
for k in 1 2 3 4 6 8; do $X --mode track --robots $k --per-call 2 --frames 400 --interval 0 --json | J; done
for k in 1 4; do $X --mode extract --robots $k --per-call 2 --frames 400 --interval 0 --json | J; done
$X --mode bf --robots 1 --json | J; $X --mode bf --robots 4 --json | J
