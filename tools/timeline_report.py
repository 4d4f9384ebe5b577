#!/usr/bin/env python3
"""Reads the rocprofv3 kernel and memory-copy traces of tools/timeline_probe.sh and prints the device timeline of the
last step: every run of same-named kernels (start, end, busy time) and every gap > 0.3 ms in which no kernel ran."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(d + "/tr/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last step = everything from the last-but-one DecodeQueueKernel's end on ... simpler: after the previous search ended
q = [i for i, r in enumerate(rows) if "DecodeQueueKernel" in r[2]]
lo = q[-2] + 1 if len(q) > 1 else 0
step = [r for r in rows[lo:] if not r[2].startswith("COPY") or True]
step = step[: [i for i, r in enumerate(step) if "DecodeQueueKernel" in r[2]][-1] + 1]
# skip what sits between the steps (host-side set-up): start at the first FeatKernel / H2D copy of the step
first = next(i for i, r in enumerate(step) if "FeatKernel" in r[2] or "COPY" in r[2] and "HOST_TO_DEVICE" in r[2].upper())
step = step[first:]
t0 = step[0][0]
print("step of %d device activities, %.2f ms from first start to last end" % (len(step), (max(r[1] for r in step) - t0) / 1e6))
run = None
busy_end = t0
for s, e, name in step:
    if name.startswith("COPY"):
        print("  %9.3f - %9.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, name))
        continue
    if s - busy_end > 0.3e6:
        print("  %9.3f - %9.3f  ---- idle %.3f ms (no kernel)" % ((busy_end - t0) / 1e6, (s - t0) / 1e6, (s - busy_end) / 1e6))
    busy_end = max(busy_end, e)
    if run and run[0] == name:
        run[2] = e; run[3] += e - s; run[4] += 1
    else:
        if run: print("  %9.3f - %9.3f  %-60s x%d busy %.3f ms" % ((run[1] - t0) / 1e6, (run[2] - t0) / 1e6, run[0], run[4], run[3] / 1e6))
        run = [name, s, e, e - s, 1]
if run: print("  %9.3f - %9.3f  %-60s x%d busy %.3f ms" % ((run[1] - t0) / 1e6, (run[2] - t0) / 1e6, run[0], run[4], run[3] / 1e6))
