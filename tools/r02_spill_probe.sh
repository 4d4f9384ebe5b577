#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the three-launch path (AdvanceKernel + FinalizeKernel2, no scratch) against the fused
# DecodeQueueKernel on the same log-likelihoods of the tglarge workload
export TMPDIR=/tmp
A="--workload librispeech --utts 256 --lanes 256 --reps 1"
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf gpurun_out/sp_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/sp_$c -o run -- python3 tools/queue_bench.py $A > gpurun_out/sp_$c.json 2> gpurun_out/sp_$c.err
  find gpurun_out/sp_$c -name "*_kernel_trace.csv" -delete
done
python3 - <<'PY'
import csv, glob, json
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    agg = {}
    for f in glob.glob("gpurun_out/sp_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0]
                a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
        print(c, k, v[0], "%.3f GB" % (v[1] * 1024 / 1e9))
try:
    print(open("gpurun_out/sp_WRITE_SIZE.json").read()[-900:])
except Exception as e:
    print(e)
PY
