"""Turns gpurun_out/round<N>/ (tools/profile_round.sh rNN on the GPU box) into the files committed under profiles/: the bench
line, the rocprofv3 --stats kernel tables (headline load; --ivectors variant) and the PMC summary bench.py reads back for
roofline.traffic -- keyed by the workload AND by the hash of the library sources it was taken with (bench.library_build_id):
a later kernel change makes bench.py report traffic null instead of these bytes.
usage: python tools/collect_profiles.py --round r06"""
import csv
import glob
import json
import os
import shutil
import sys

_args = [x for x in sys.argv[1:] if x != "--round"]
tag = _args[0] if _args else "r06"
if not (len(tag) == 3 and tag[0] == "r" and tag[1:].isdigit()):
    sys.exit("usage: python tools/collect_profiles.py --round rNN")
R = "gpurun_out/round%d" % int(tag[1:])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (library_build_id)
P = "profiles"
os.makedirs(P, exist_ok=True)


def line(path):
    if not os.path.exists(path):
        return None
    for l in open(path):
        if l.startswith("{"):
            return json.loads(l)
    return None


def rows(dirname):
    for f in glob.glob(os.path.join(R, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            yield row


def per_kernel(dirname, name, kernel):
    return [float(r["Counter_Value"]) for r in rows(dirname) if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]


base = line(os.path.join(R, "bench_default.json"))
if base:
    json.dump(base, open(os.path.join(P, "%s_bench_default.json" % tag), "w"), indent=1)
d = line(os.path.join(R, "bench_under_rocprof.json"))
if d:
    json.dump(d, open(os.path.join(P, "%s_bench_under_rocprof.json" % tag), "w"), indent=1)
for f in glob.glob(os.path.join(R, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "%s_kernel_stats_bench_default.csv" % tag))
for f in glob.glob(os.path.join(R, "stats_random", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "%s_kernel_stats_bench_random_headline.csv" % tag))
div = line(os.path.join(R, "bench_random_under_rocprof.json"))
if div:
    json.dump(div, open(os.path.join(P, "%s_bench_random_under_rocprof.json" % tag), "w"), indent=1)
cfg = (base or d or {}).get("config", {})
prof = d or line(os.path.join(R, "pmc_fetch.json")) or {}
# bench steps a profiled pass runs: warmup + steps of the headline, and 1 + min(3, steps) of the resident leg when it ran
STEPS_PER_PASS = float(prof.get("warmup", 1) + prof.get("steps", 2) + ((1 + min(3, prof.get("steps", 2))) if "hbm_resident_value" in prof else 0))
out = {"round": tag, "device": "MI355X (gfx950), ROCm 7.2",
       "command": "rocprofv3 --kernel-trace --pmc <counter set> (one set per pass) -- python3 bench.py --steps 2 --warmup 1 "
                  "--no-random-leg --no-cpu-baseline --no-wer --no-streaming",
       "library_build": bench.library_build_id(),
       "note": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): read bytes = "
               "2 * FETCH_SIZE * 1024 (an upper bound for the decoder's narrow random reads), written bytes = WRITE_SIZE * 1024.  "
               "Per pass: warmup + timed steps with the upload in the timed region, then 1 + min(3, steps) with the waveforms resident (the hbm_resident leg): sums over all launches divided by bench_steps_per_profiled_pass.",
       "workload": cfg.get("workload")}
if base:
    out["workload_key"] = "librispeech/tglarge/%d/faithful/%s/%s/%s" % (cfg.get("utterances", 0), cfg.get("planted_peak"), cfg.get("planted_noise"), cfg.get("lm_scale"))
out["bench_steps_per_profiled_pass"] = STEPS_PER_PASS
for kern, key in (("DecodeQueueKernel", "decode_queue"), ("TdnnGemm", "gemm_all_layers"), ("FeatKernel", "features")):
    fe, wr = per_kernel("pmc_fetch", "FETCH_SIZE", kern), per_kernel("pmc_write", "WRITE_SIZE", kern)
    if fe and wr:
        n = STEPS_PER_PASS                                        # per bench step: all launches of the kernel family over the pass's steps
        out[key] = {"launches_in_pass": len(fe), "FETCH_SIZE_KB_per_step": sum(fe) / n, "WRITE_SIZE_KB_per_step": sum(wr) / n,
                    "traffic_bytes_per_step": 2 * 1024 * sum(fe) / n + 1024 * sum(wr) / n}
for kern, key in (("DecodeQueueKernel", "decode_queue"), ("TdnnGemm", "gemm_all_layers")):
    sq = {}
    for dname in ("pmc_sq1", "pmc_sq3"):
        for r in rows(dname):
            if kern in r["Kernel_Name"]:
                sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if sq:
        s = {k: sum(v) / STEPS_PER_PASS for k, v in sq.items()}
        out[key + "_sq_counters_per_step"] = s
        if s.get("SQ_WAVE_CYCLES", 0) > 0:
            out[key + "_wait_fraction"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
        if s.get("TCC_HIT_sum", 0) + s.get("TCC_MISS_sum", 0) > 0:
            out[key + "_l2_hit_rate"] = s["TCC_HIT_sum"] / (s["TCC_HIT_sum"] + s["TCC_MISS_sum"])
mf = {}
for r in rows("pmc_mfma"):
    if r["Counter_Name"] == "MfmaUtil" and "TdnnGemm" in r["Kernel_Name"]:
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        k = r["Kernel_Name"].split("(")[0].replace("void kamd::", "")
        a = mf.setdefault(k, [0.0, 0.0, 0])
        a[0] += float(r["Counter_Value"]) * dur; a[1] += dur; a[2] += 1
if mf:
    out["mfma_util_percent"] = {k: {"launches": v[2], "time_weighted_MfmaUtil": v[0] / max(v[1], 1.0), "total_ms": v[1] / 1e6}
                                for k, v in mf.items()}
if "decode_queue" in out:
    out["kernel"] = "kamd::DecodeQueueKernel"
    out["traffic_bytes_per_launch"] = out["decode_queue"]["traffic_bytes_per_step"]
    for src in (base, d):
        if src:
            r = src["roofline"] if src["roofline"]["bound"] == "hbm" else src["roofline_other_stage"]
            out["algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
            break
json.dump(out, open(os.path.join(P, "%s_pmc.json" % tag), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
