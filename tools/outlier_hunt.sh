for i in $(seq 1 40); do python bench.py --no-cpu-baseline --no-replay --steps 10 2>/dev/null | tail -1 > /tmp/o_$i.json; python - <<PY
import json
d=json.load(open("/tmp/o_$i.json"))
v=d["value"]
print($i, round(v), {k: round(x,3) for k,x in d["roofline"]["kernel_ms_per_step"].items()} if v < 90000 else "")
PY
done
