ORBSLAMM_HIP_LIB=$PWD/build_ub/libB.so python tools/orient_timing.py 2>&1 | grep ORIENT > gpurun_out/orient_t.txt; python - <<PY
import re
import numpy as np
rows=[]
for l in open("gpurun_out/orient_t.txt"):
    m=re.findall(r"([a-z]+) ([0-9.]+)", l.split(":")[1])
    rows.append([float(v) for k,v in m])
a=np.array(rows); print(len(a)); print("barrier counts rec moments trig brief"); print(a.mean(0)); print(np.median(a,0))
PY
