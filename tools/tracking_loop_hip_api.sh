# on the GPU box: the HIP-API trace of examples/tracking_loop (the drop-in classes in Tracking's shape) at two lengths.
# A steady-state loop makes no allocation, stream or event: the counts of those APIs must not grow with the frame count.
# usage: bash tools/tracking_loop_hip_api.sh > gpurun_out/tracking_loop_hip_api.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for n in 100 400; do
    rm -rf /tmp/tl_api_$n
    timeout 300 rocprofv3 --hip-runtime-trace --stats -f csv -d /tmp/tl_api_$n -o t -- $R/examples/tracking_loop --frames $n --warmup 20 --no-old-pattern > /tmp/tl_api_$n.out 2>&1
    tail -12 /tmp/tl_api_$n.out | head -9
done
python3 $R/tools/hip_api_diff.py /tmp/tl_api_100 /tmp/tl_api_400 300
