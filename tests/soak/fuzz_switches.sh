#!/bin/bash
# every run-time switch the library still reads (six since round 6: docs/experiments.md) must leave every result byte alone: the extractor + stream soaks under
# each of them.  bash tests/soak/fuzz_switches.sh [cases] > gpurun_out/fuzz_switches.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-150}
run() { # NAME=VALUE ...
  out=$(env "$@" python $R/tests/soak/fuzz_soak.py $N 61 2>&1 | grep -E "^fuzz soak|DIFFERENCE|ERROR|Error" | head -2)
  out2=$(env "$@" python $R/tests/soak/fuzz_stream.py 200 62 2>&1 | grep -E "^stream soak|DIFFERENCE|ERROR|Error" | head -2)
  echo "$* :: ${out:0:110} :: ${out2:0:90}"
}
run ORBX_NONE=1
run ORBX_MATCH_POPCOUNT=1
run ORBX_SERIAL=1
run ORBX_LAT_STREAMS=2
run ORBX_LAT_PRIO=0
run ORBX_STAGE_NT=0
run ORBX_COPY_THREADS=2
