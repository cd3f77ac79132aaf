import sys, numpy as np
sys.path.insert(0, '.')
from orbslamm_amd import ORBextractor, synth
B=64
fr = synth.make_frames(1241, 376, B)
ex = ORBextractor(2000, 1.2, 8, 20, 7, max_width=1241, max_height=376, max_batch=B, device=0)
ex.set_serial(1)
d = ex.upload_frames(fr, stride=1280)
for i in range(2):
    ex.extract_batch_device(*d); ex.sync()
