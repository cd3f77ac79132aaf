#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_run.sh <tag> "<counters>" [bench args]
# one rocprofv3 --pmc pass of a short bench run; summary printed and left in gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc $CTR -d $R/gpurun_out/pmc_$TAG -o p -- python $R/bench.py --no-cpu-baseline --no-profile --steps 5 --warmup 2 "$@" > /dev/null 2> $R/gpurun_out/pmc_$TAG.err
python $R/tools/pmc_table.py $R/gpurun_out/pmc_$TAG/p_results.db | tee $R/gpurun_out/pmc_$TAG.txt
