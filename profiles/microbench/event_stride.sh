for rep in 1 2 3; do for st in 4 8 16 0; do
  if [ $st = 0 ]; then extra="--no-profile"; else extra="--event-stride $st"; fi
  python3 bench.py --no-cpu-baseline --no-serving --no-host-to-host --no-large-batch --no-single-solve --no-reference-faithful $extra 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}; print('stride $st', round(j['value']), round(j['ms_per_step'],4), r.get('avg_launch_us'), r.get('timed_launches'))"
done; done
