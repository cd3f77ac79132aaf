for i in 1 2 3; do
for a in "" "--no-profile"; do
python bench.py --no-cpu-baseline --no-replay $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$a]', round(d['value']), round(d['ms_per_step'],4))"
done; done
