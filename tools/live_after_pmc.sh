#!/bin/bash
# on the GPU box: the first live-stream run after a rocprofv3 --pmc pass (bench.py's live_streams timed out there once)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
cd /tmp && export TMPDIR=/tmp
pmc() { ORBX_SERIAL=1 timeout 240 rocprofv3 --pmc $1 -d /tmp/pmc_x -o p -- python $R/bench.py --no-cpu-baseline --no-profile --no-host-path --no-tracking-path --no-parity-check --pool 2 --steps 5 --warmup 2 > /dev/null 2>&1; echo "pmc pass rc=$?"; }
for fp in 1 0 1; do
  pmc "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE"
  SECONDS=0
  ORBX_FUSE_PACK=$fp timeout 100 $X --json --interval 0 --w 1241 --h 376 --nfeat 2000 --gpus 1 --frames 600 --warmup 40 --mode track 2>&1 | tail -1 | cut -c1-200
  echo "fuse=$fp rc=$? took $SECONDS s"
done
