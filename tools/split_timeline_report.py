#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace of tools/shard_probe.py (a rank's shard with the long-utterance decoder) and prints
the last step's timeline with the hardware queue of every kernel: the two work-queue kernels, the feature and descriptor
kernels, every kernel longer than 1.5 ms, and the span the GEMMs cover.  Two streams that show the same Queue_Id run
their kernels in submission order (profiles/r03_timeline_upload.txt, last block).

  rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 tools/shard_probe.py --worlds 8 --long-lanes 32 --steps 1
  python3 tools/split_timeline_report.py DIR"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")))
rows.sort()
q = [i for i, r in enumerate(rows) if "DecodeQueueKernel" in r[2]]
if len(q) < 2:
    sys.exit("no step with two work-queue kernels in the trace")
i0 = max(i for i in range(q[-2]) if "FeatKernel" in rows[i][2])     # the last step starts at its feature kernel
t0 = rows[i0][0]
for s, e, n, qid in rows[i0:q[-1] + 1]:
    if "Decode" in n or "Feat" in n or "Pull" in n or (e - s) > 1.5e6:
        print("%9.3f - %9.3f q%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, qid, n))
g = [(s, e) for s, e, n, qid in rows[i0:q[-1] + 1] if "Gemm" in n]
if g:
    print("gemm kernels", len(g), "busy %.2f ms" % (sum(e - s for s, e in g) / 1e6), "span %.2f-%.2f" % ((g[0][0] - t0) / 1e6, (g[-1][1] - t0) / 1e6))
