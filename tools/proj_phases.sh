#!/bin/bash
# here: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DORBT_PHASE_TIMING -o build_ub/libT.so orbslamm_amd/csrc/orbslamm_hip.hip
# gpurun: bash tools/proj_phases.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
ORBSLAMM_HIP_LIB=$R/build_ub/libT.so python $R/tools/proj_phases.py
ORBSLAMM_HIP_LIB=$R/build_ub/libT.so python $R/tools/bow_phases.py
