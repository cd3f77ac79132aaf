X=./examples/multi_robot
J() { grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-8s thr %2d x %d pinned %d %8.0f frames/s  median %.4f mean %.4f p99 %.4f' % (d['mode'], d['robots'], d['cameras_per_call'], d['pinned'], d['frames_per_s'], d['ms_median'], d['ms_mean'], d['ms_p99']))"; }
for dma in 1 0 1 0; do
  export ORBX_LAT_DMA=$dma; echo "== ORBX_LAT_DMA=$dma"
  $X --mode track --interval 0 --json | J
  $X --mode track --pinned 0 --interval 0 --json | J
  $X --mode bf --interval 0 --json | J
  $X --mode track --robots 4 --frames 400 --interval 0 --json | J
  $X --mode track --robots 4 --per-call 2 --frames 400 --interval 0 --json | J
done
unset ORBX_LAT_DMA
bash tools/live_dma_gap.sh track | head -14
