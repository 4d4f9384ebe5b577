import json
import sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print("[%s] RTF %.0f  ms/step %.1f  stages %s" % (tag, d["value"], d["ms_per_step"],
              {k: round(v, 2) for k, v in d["stage_ms"].items()}))
        print("    decoder %s" % {k: round(v, 1) for k, v in d["decoder"].items()})
        if "phase_share_longest_lane" in d:
            print("    phases %s" % d["phase_share_longest_lane"])
        r = d["roofline"]
        print("    roofline %.1f GB/s (%.4f of peak), nnet %.1f TFLOP/s, cpu %s" % (
            r["achieved"], r["frac"], d["nnet_tflops"], d.get("cpu_baseline") and round(d["cpu_baseline"]["value"], 2)))
    else:
        print(l[:300].rstrip())
