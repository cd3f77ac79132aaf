import sys, numpy as np
sys.path.insert(0, '.')
from orbslamm_amd import ORBextractor, synth
fr = synth.make_frames(1241, 376, 2)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=2, device=0)
ex.set_serial(1)
for i in range(2):
    k, d = ex.extract_batch(fr)
print(len(k[0]))
