# phase timing of k_distribute (level 0 of frame 0): build the library with -DORBX_DIST_TIMING first, e.g.
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DORBX_DIST_TIMING -o orbslamm_amd/liborbslamm_hip.so orbslamm_amd/csrc/orbslamm_hip.hip
import sys, numpy as np
sys.path.insert(0, '.')
from orbslamm_amd import ORBextractor, synth
import os
NB = int(os.environ.get("DIST_FRAMES", "2"))
fr = synth.make_frames(1241, 376, NB)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=NB, device=0)
ex.set_serial(1)
for i in range(2):
    k, d = ex.extract_batch(fr)
print(len(k[0]))
