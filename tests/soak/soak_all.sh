#!/bin/bash
# every differential soak with fresh seeds, one gpurun call: bash tests/soak/soak_all.sh SEED > gpurun_out/soak_all.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-1000}
cd $R
python tests/soak/fuzz_soak.py 2500 $S 2>&1 | grep -E "^fuzz soak|DIFFERENCE|ERROR|Error"
python tests/soak/fuzz_matchers.py 500 $((S+1)) 2>&1 | grep -E "^matcher soak|DIFFERENCE|ERROR|Error"
python tests/soak/fuzz_stream.py 4000 $((S+2)) 2>&1 | grep -E "^stream soak|DIFFERENCE|ERROR|Error"
python tests/soak/fuzz_frontend.py 500 $((S+3)) 3000 2>&1 | grep -E "^front-end soak|DIFFERENCE|ERROR|Error"
python tests/soak/fuzz_tracking.py 1500 $((S+4)) 3000 2>&1 | grep -E "^tracking soak|DIFFERENCE|ERROR|Error"
bash tests/soak/fuzz_dropin.sh 150 2>&1 | tail -1
