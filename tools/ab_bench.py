#!/usr/bin/env python3
"""A/B of library builds on the headline workload, on the same box, back to back.

    python tools/ab_bench.py name=path/to/lib.so [name2=...] [-- extra bench.py flags]

Every build runs `bench.py --steps 3 --warmup 1` (the recipe-faithful headline and the random-log-likelihood leg; no CPU
baseline, WER or streaming legs) under KAMD_LIB=<path>; the JSON lines land in gpurun_out/ab/<name>.json and one summary
line per build is printed: ms per step, the stages, search microseconds per frame and lane, the leg's rate.  A name may
carry environment settings: name:VAR=value,VAR2=value=path (experiments behind KAMD_* switches).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    pairs = [a.rsplit("=", 1) for a in argv if "=" in a]
    out = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out, exist_ok=True)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-wer", "--no-streaming", "--no-planted", "--no-ivector-leg"]
    for name, lib in pairs:
        env = dict(os.environ, KAMD_LIB=os.path.abspath(lib))
        if ":" in name:
            name, settings = name.split(":", 1)
            for kv in settings.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        r = subprocess.run(base + extra, cwd=ROOT, env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED rc", r.returncode, r.stderr[-1500:], flush=True)
            continue
        with open(os.path.join(out, name + ".json"), "w") as f:
            f.write(line[-1] + "\n")
        j = json.loads(line[-1])
        msg = "%-10s %8.1f ms/step  x_rt %8.0f  stage_ms %s  search us/frame/lane %.1f" % (
            name, j["ms_per_step"], j["value"], json.dumps(j.get("stage_ms")), j["roofline"].get("us_per_frame_per_lane", -1))
        hbm = j["roofline"] if j["roofline"].get("bound") == "hbm" else j.get("roofline_other_stage", {})
        msg = "%-10s %8.1f ms/step  x_rt %8.0f  stage_ms %s  search us/frame/lane %.1f failed %s retried %s" % (
            name, j["ms_per_step"], j["value"], json.dumps({k: round(v, 1) for k, v in j.get("stage_ms", {}).items()}),
            hbm.get("us_per_frame_per_lane", -1), j["decoder"].get("failed_utterances"), j.get("n_retried"))
        for leg in ("random_loglikes", "planted", "online_ivectors"):
            if isinstance(j.get(leg), dict) and "value" in j[leg]:
                msg += "  | %s %.0f (search %.1f ms)" % (leg, j[leg]["value"], j[leg].get("stage_ms", {}).get("decode_queue_kernel", -1))
        print(msg, flush=True)


if __name__ == "__main__":
    main()
