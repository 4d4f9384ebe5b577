"""Turns gpurun_out/round/ (written by tools/profile_round.sh on the GPU box) into the files
committed under profiles/: the bench lines, the rocprofv3 --stats kernel table and the PMC
summary bench.py reads back for roofline.traffic.  usage: python tools/collect_profiles.py r01"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = "gpurun_out/round"
P = "profiles"
os.makedirs(P, exist_ok=True)


def line(path):
    for l in open(path):
        if l.startswith("{"):
            return json.loads(l)
    return None


def counter(dirname, name, kernel):
    vals = []
    for f in glob.glob(os.path.join(R, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name and kernel in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return vals


for name in ("default", "light", "saturated", "b256", "b512", "ivectors"):
    d = line(os.path.join(R, "bench_%s.json" % name))
    if d:
        json.dump(d, open(os.path.join(P, "%s_bench_%s.json" % (tag, name)), "w"), indent=1)
d = line(os.path.join(R, "bench_under_rocprof.json"))
if d:
    json.dump(d, open(os.path.join(P, "%s_bench_under_rocprof.json" % tag), "w"), indent=1)
for f in glob.glob(os.path.join(R, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "%s_kernel_stats_bench_default.csv" % tag))
for f in glob.glob(os.path.join(R, "stats_iv", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "%s_kernel_stats_bench_ivectors.csv" % tag))
base = line(os.path.join(R, "bench_default.json"))
out = {"round": tag, "device": "MI355X (gfx950), ROCm 7.2",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                  "--warmup 1 --no-cpu-baseline",
       "note": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); "
               "read bytes = 2*FETCH_SIZE*1024 is therefore an upper bound for this kernel's narrow random reads.",
       "workload_key": "mini_librispeech/64/1.3/0.1"}
for kern, key in (("AdvanceKernel", "advance"), ("FinalizeKernel", "finalize")):
    fe, wr = counter("pmc_fetch", "FETCH_SIZE", kern), counter("pmc_write", "WRITE_SIZE", kern)
    if fe and wr:
        out[key] = {"launches": len(fe), "FETCH_SIZE_KB_mean": sum(fe) / len(fe), "WRITE_SIZE_KB_mean": sum(wr) / len(wr),
                    "traffic_bytes_per_launch": 2 * 1024 * sum(fe) / len(fe) + 1024 * sum(wr) / len(wr)}
sq = {}
for dname in ("pmc_sq1", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob(os.path.join(R, dname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "AdvanceKernel" in row["Kernel_Name"]:
                sq.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if sq:
    out["advance_sq_counters_per_launch"] = {k: sum(v) / len(v) for k, v in sq.items()}
    w = out["advance_sq_counters_per_launch"]
    if "SQ_WAVE_CYCLES" in w and "SQ_WAIT_ANY" in w and w["SQ_WAVE_CYCLES"] > 0:
        out["advance_wait_fraction"] = w["SQ_WAIT_ANY"] / w["SQ_WAVE_CYCLES"]
mf = {}
for f in glob.glob(os.path.join(R, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "MfmaUtil" and "TdnnGemmKernel" in row["Kernel_Name"]:
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            k = row["Kernel_Name"].split("(")[0].replace("void kamd::", "")
            a = mf.setdefault(k, [0.0, 0.0, 0])
            a[0] += float(row["Counter_Value"]) * dur; a[1] += dur; a[2] += 1
if mf:
    out["mfma_util_percent"] = {k: {"launches": v[2], "time_weighted_MfmaUtil": v[0] / max(v[1], 1.0)} for k, v in mf.items()}
    out["mfma_note"] = ("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100, own pass "
                        "(rocprofv3 --pmc MfmaUtil); fp32 MFMA peak 157 TFLOP/s")
if "advance" in out:
    out["kernel"] = "kamd::AdvanceKernel"
    out["traffic_bytes_per_launch"] = out["advance"]["traffic_bytes_per_launch"]
    if base:
        out["algorithmic_bytes_per_launch"] = base["roofline"]["algorithmic_bytes_per_launch"]
    json.dump(out, open(os.path.join(P, "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
