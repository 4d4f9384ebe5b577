cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python3 bench.py --ivectors --steps 3 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rtf %.0f ms/step %.2f' % (d['value'], d['ms_per_step']), d['stage_ms'], d['stage_ms_unsliced'])
"
mkdir -p gpurun_out/iv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/iv -- python3 bench.py --ivectors --steps 2 --warmup 1 > gpurun_out/iv/bench.log 2>&1
f=$(find gpurun_out/iv -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
