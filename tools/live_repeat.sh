#!/bin/bash
# on the GPU box: run-to-run spread of the K-robot configurations (queue assignment at handle creation)
X=${GRAFT_REPO_ROOT:-/root/repo}/examples/multi_robot
for i in $(seq ${1:-8}); do
  a=$($X --mode track --robots 4 --per-call 2 --frames 400 --interval 0 --json | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']))")
  b=$($X --mode track --robots 4 --frames 400 --interval 0 --json | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']))")
  c=$($X --mode bf --robots 4 --frames 400 --interval 0 --json | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']))")
  e=$($X --mode track --robots 8 --frames 400 --interval 0 --json | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']))")
  echo "4x2 $a   4x1 $b   bf4 $c   8x1 $e"
done
