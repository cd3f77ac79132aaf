#!/bin/bash
# on the GPU box: tickets two deep, extraction on the extractor's queue beside the search on the matcher's -- timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
$R/examples/multi_robot --mode track --depth 2 --attach 0 --interval 0 --json | grep '^{' | cut -c1-300
GPU_MAX_HW_QUEUES=8 $R/examples/multi_robot --mode track --depth 2 --attach 0 --interval 0 --json | grep '^{' | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/d2prof -o t -- $R/examples/multi_robot --mode track --depth 2 --attach 0 --frames 60 --warmup 10 --interval 0 --json > /dev/null 2>&1
python $R/tools/timeline.py /tmp/d2prof/t_results.db 40 20
