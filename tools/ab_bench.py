#!/usr/bin/env python3
"""A/B of library builds on the headline workload, on the same box, back to back.

    python tools/ab_bench.py [--planted] [--ivectors] name=path/to/lib.so [name2=...] [-- extra bench.py flags]

Every build runs `bench.py --steps 3 --warmup 1` with the slow legs off (no bracket, no cpu baseline, no WER; the
planted and i-vector legs only when asked) under KAMD_LIB=<path>; the JSON lines land in gpurun_out/ab/<name>.json and
one summary line per build is printed: ms per step, the stages, search microseconds per frame and lane.
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
    planted = "--planted" in argv
    ivectors = "--ivectors" in argv
    pairs = [a.split("=", 1) for a in argv if "=" in a]
    out = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out, exist_ok=True)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-bracket", "--no-cpu-baseline", "--no-wer", "--no-streaming"]
    if not planted:
        base.append("--no-planted")
    if not ivectors:
        base.append("--no-ivector-leg")
    for name, lib in pairs:
        env = dict(os.environ, KAMD_LIB=os.path.abspath(lib))
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
        for leg in ("planted", "online_ivectors"):
            if isinstance(j.get(leg), dict) and "value" in j[leg]:
                msg += "  %s %.0f (failed %s)" % (leg, j[leg]["value"], j[leg].get("failed_utterances"))
        print(msg, flush=True)


if __name__ == "__main__":
    main()
