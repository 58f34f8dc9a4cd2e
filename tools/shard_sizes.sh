for n in 500000 250000 125000; do
  python bench.py --cells $n --no-cpu-baseline --no-host-delivery --steps 3 2>/dev/null > /tmp/s_$n.json
  python -c "
import json
d=json.loads(open('/tmp/s_$n.json').read().strip().splitlines()[-1]); print($n, d['ms_per_step'], d['value'])"
done
