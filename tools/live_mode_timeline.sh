#!/bin/bash
# on the GPU box: kernel timeline of one robot's chain in a given mode (track | bf | extract | full)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-bf track}; do
  rm -rf /tmp/modeprof
  timeout 200 rocprofv3 --kernel-trace -d /tmp/modeprof -o t -- $X --mode $m --robots 1 --frames 60 --warmup 20 --interval 0 --json > /dev/null 2>&1
  echo "== mode $m"
  python3 $R/tools/timeline.py /tmp/modeprof/t_results.db ${NK:-14} 28
done
