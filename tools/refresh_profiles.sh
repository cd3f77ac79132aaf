#!/bin/bash
# On the GPU box (via gpurun): regenerate everything kept under profiles/ for round TAG (default r01) into
# gpurun_out/.  rocprofv3 passes are separate runs: --kernel-trace --stats, then one --pmc pass per counter set
# (never combined with tracing), PMC passes in ORBX_SERIAL=1 so that every kernel is alone on the GPU.  Every profiler
# run sits under its own `timeout`: a pass with derived TA_*/TCP_* counters once never returned and cost 10 GPU-minutes.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# the counter passes FIRST: bench.py replays their tables (profiles/${TAG}_pmc_*.json) and flags them stale when the kernel sources changed since
export ORBX_SERIAL=1
pmc() { # tag, counters
  timeout 240 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o p -- python $R/bench.py --no-cpu-baseline --no-profile --no-host-path --no-tracking-path --no-natural --no-parity-check --pool 2 --steps 5 --warmup 2 > /dev/null 2>&1
}
pmc fetch "FETCH_SIZE"; python $R/tools/rocprof_summary.py pmc /tmp/pmc_fetch/p_results.db > $O/${TAG}_pmc_fetch.txt
pmc write "WRITE_SIZE"; python $R/tools/rocprof_summary.py pmc /tmp/pmc_write/p_results.db > $O/${TAG}_pmc_write.txt
python $R/tools/pmc_traffic.py /tmp/pmc_fetch/p_results.db /tmp/pmc_write/p_results.db $O/${TAG}_pmc_traffic.json > /dev/null
pmc sqa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; python $R/tools/pmc_table.py /tmp/pmc_sqa/p_results.db > $O/${TAG}_pmc_sq_a.txt
pmc sqb "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; python $R/tools/pmc_table.py /tmp/pmc_sqb/p_results.db > $O/${TAG}_pmc_sq_b.txt
pmc sqc "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE"; python $R/tools/pmc_table.py /tmp/pmc_sqc/p_results.db > $O/${TAG}_pmc_sq_c.txt
python $R/tools/pmc_sq_json.py $O/${TAG}_pmc_sq.json /tmp/pmc_sqa/p_results.db /tmp/pmc_sqb/p_results.db /tmp/pmc_sqc/p_results.db > /dev/null
unset ORBX_SERIAL
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_sq.json $R/profiles/
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-tracking-path > $O/${TAG}_bench_20steps.json 2>> $O/${TAG}_bench.err
python $R/bench.py --config c2 --no-cpu-baseline --no-host-path --no-tracking-path > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench.err
ORBX_SERIAL=1 python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path > $O/${TAG}_bench_serial.json 2>> $O/${TAG}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o t -- python $R/bench.py --no-cpu-baseline --no-replay --no-host-path --no-tracking-path --no-parity-check > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py stats /tmp/prof_o/t_results.db > $O/${TAG}_kernel_stats.txt
python $R/tools/timeline.py /tmp/prof_o/t_results.db 70 300 > $O/${TAG}_timeline.txt
ORBX_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o t -- python $R/bench.py --no-cpu-baseline --no-replay --no-host-path --no-tracking-path --no-parity-check > /dev/null 2>&1
python $R/tools/rocprof_summary.py stats /tmp/prof_s/t_results.db > $O/${TAG}_kernel_stats_serial.txt
if [ -x $R/build_ub/fetch_calib ]; then
  rocprofv3 --pmc FETCH_SIZE -d /tmp/c1 -o p -- $R/build_ub/fetch_calib > /dev/null 2>&1; rocprofv3 --pmc WRITE_SIZE -d /tmp/c2 -o p -- $R/build_ub/fetch_calib > /dev/null 2>&1
  { echo "# tools/ubench/fetch_calib.hip: read_b1 = 64 MiB at 1 B/lane, read_b4 / read_b16 = 512 MiB at 4 / 16 B/lane, write_b4 = 512 MiB (values in KB)"; python $R/tools/rocprof_summary.py pmc /tmp/c1/p_results.db; python $R/tools/rocprof_summary.py pmc /tmp/c2/p_results.db; } > $O/${TAG}_fetch_calibration.txt
fi
# the Tracking-shaped path: bench.py's tracking_path block alone, then its kernels under rocprofv3 (B = 1 and B = 64 launches mixed:
# the per-kernel averages are over both; tools/proj_phases.sh has the per-phase times of one resolve workgroup)
python $R/tools/tracking_bench.py > $O/${TAG}_tracking_path.json 2>> $O/${TAG}_bench.err
TRK_SECONDS=0.3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/tools/tracking_bench.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py stats /tmp/prof_t/t_results.db > $O/${TAG}_tracking_kernel_stats.txt
ls -la $O | tail -15
