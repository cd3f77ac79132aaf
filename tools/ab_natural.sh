R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then E=""; else E="$cfg"; fi
    env $E python $R/bench.py --no-cpu-baseline --no-host-path --no-tracking-path --no-live-streams --no-dropin-classes 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['isolated']['kernel_ms_per_step']; n=d.get('natural',{})
print('[$cfg]', round(d['value']), 'parity', d.get('parity_check',{}).get('ok'), 'fast', round(k['k_fast'],4), {q: (round(n[q]['fps']), n[q]['parity_ok']) for q in ('retina_pan','mosaic','hubble') if q in n})"
  done
done
