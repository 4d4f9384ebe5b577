set -x
python bench.py --workload tiny --verbose > gpurun_out/r02_tiny.json 2> gpurun_out/r02_tiny.err; tail -5 gpurun_out/r02_tiny.err; cat gpurun_out/r02_tiny.json | head -c 3000; echo
python bench.py --workload tiny --gpus 2 --dist-backend gloo --device 0 --no-wer > gpurun_out/r02_tiny_g2.json 2> gpurun_out/r02_tiny_g2.err; tail -5 gpurun_out/r02_tiny_g2.err; cat gpurun_out/r02_tiny_g2.json | head -c 1500; echo
for cfgs in "0.3 1.6" "0.2 1.6" "0.3 1.3" "0.5 1.3"; do set -- $cfgs; python bench.py --utts 320 --steps 1 --warmup 1 --no-wer --no-cpu-baseline --verbose --lm-scale $1 --ll-std $2 > gpurun_out/r02_cal_$1_$2.json 2> gpurun_out/r02_cal_$1_$2.err; tail -4 gpurun_out/r02_cal_$1_$2.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r02_cal_$1_$2.json')); print('$1 $2', d['value'], d['stage_ms'], d['decoder'], d['roofline']['frac'], d['roofline_other_stage']['achieved'])"; done
