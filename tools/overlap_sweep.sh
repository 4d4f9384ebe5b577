for ov in "" "32,96" "24,72,216" "48" "64,192" "16,48,144,432"; do
  timeout 300 python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 --overlap "$ov" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ov=%-16s' % '$ov', 'rtf %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['stage_ms'].items()}, 'launch_ms %.2f x%d frac %.4f' % (d['roofline']['launch_ms'], d['roofline']['launches_per_step'], d['roofline']['frac']))
"
done
