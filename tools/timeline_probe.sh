#!/bin/bash
# One headline step under rocprofv3's kernel + memory-copy trace with the batch decoder's host marks (KAMD_BATCH_TRACE):
# where the wall time of a step goes that the stage events do not cover.  usage (gpurun): tools/timeline_probe.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/timeline
rm -rf $O; mkdir -p $O
H="--no-bracket --no-planted --no-ivector-leg --no-cpu-baseline --no-wer"
export KAMD_BATCH_TRACE=1
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o run -- python3 bench.py --steps 1 --warmup 1 $H > $O/bench.json 2> $O/bench.err
python3 tools/timeline_report.py $O > $O/report.txt 2>&1
find $O -name "*.csv" -size +20M -delete
tail -60 $O/report.txt
