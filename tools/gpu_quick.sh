# on the GPU box: full GPU suite + one bench line (isolated kernel times)
timeout 400 python -m pytest tests -q -m gpu --timeout=300 -x 2>&1 | tail -3
python bench.py > gpurun_out/b.json 2>gpurun_out/b.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["matches_last_frame"]); print(d["roofline"]["isolated"]["kernel_ms_per_step"])
PY
