#!/bin/bash
# A/B of the upload of orbm_track_local_points' query block: the copy kernel of the live chain against the DMA engine
# (ORBM_Q_DMA=1).  Reads tracking_path.local_map / projected_pose of bench.py, three rounds each, alternating.
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in 0 1; do
    ORBM_Q_DMA=$v timeout 300 python bench.py --steps 20 --no-live-streams --no-host-path --no-cpu-baseline --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['tracking_path']
lm=d['local_map']
print('dma=$v', 'dropin %.4f' % d['dropin']['ms_median'], d['parity_ok'], 'th1 %.4f th3 %.4f pipelined %.4f/%.4f parity %s %s | projected %.4f %s' % (lm['th1']['ms_median'], lm['th3']['ms_median'], lm['th1']['pipelined_ms_per_call'], lm['th3']['pipelined_ms_per_call'], lm['th1']['parity_ok'], lm['th3']['parity_ok'], d['projected_pose']['ms_median'], d['projected_pose']['parity_ok']))"
  done
done
