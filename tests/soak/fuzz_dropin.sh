#!/bin/bash
# the drop-in ORBmatcher classes (include/ORBmatcher_hip.hpp, all eleven members on mock SLAM objects, each call checked against
# the C oracle inside tests/cpp/matcher_dropin_gpu.cpp) over many seeds of the synthetic world: bash tests/soak/fuzz_dropin.sh [seeds]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-40}
EXE=/tmp/matcher_dropin_gpu
g++ -std=c++11 -O1 -Wall -Werror -I $R/include $R/tests/cpp/matcher_dropin_gpu.cpp -o $EXE -L $R/orbslamm_amd -lorbslamm_hip -L $R/oracle -lorb_oracle \
    -Wl,-rpath,$R/orbslamm_amd -Wl,-rpath,$R/oracle -Wl,-rpath,/opt/rocm/lib || exit 2
ok=0
for s in $(seq 1 $N); do
  out=$($EXE $s $((600 + (s * 37) % 1600)) 2>&1)   # world sizes 600 .. 2199 MapPoints
  if echo "$out" | grep -q "matcher_dropin_gpu ok"; then ok=$((ok+1)); else echo "seed $s FAILED:"; echo "$out" | tail -15; exit 1; fi
done
echo "drop-in classes: $ok of $N seeds, every member call equal to the oracle's and every write-back as the reference's"
