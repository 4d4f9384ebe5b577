# the i-vector variant of the bench (secondary measurement): --ll-std is raised until the
# search load (expanded tokens per frame) matches the headline run's
for std in 1.3 1.8 2.3 3.0; do
timeout 400 python3 bench.py --ivectors --ll-std $std --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ll-std $std rtf %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), {k: (round(v, 2) if v is not None else None) for k, v in d['stage_ms'].items()}, 'expanded/frame %.0f arcs/frame %.0f' % (d['decoder']['expanded_per_frame'], d['decoder']['arcs_per_frame']))
"
done
