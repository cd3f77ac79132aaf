R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export ORBX_SERIAL=1
pmc() { timeout 240 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py --no-cpu-baseline --no-profile --no-host-path --no-tracking-path --no-parity-check --pool 2 --steps 5 --warmup 2 > /dev/null 2>&1; python $R/tools/pmc_table.py /tmp/pmc_$1/p_results.db | head -8; }
pmc a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
pmc b "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
pmc c "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
