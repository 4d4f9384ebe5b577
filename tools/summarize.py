"""One-line summary of bench.py JSON lines.
  python bench.py ... | python tools/summarize.py [tag]      or      python tools/summarize.py tag file.json [file2.json ...]
With file arguments nothing is read from stdin (a call without a pipe must not sit waiting for input)."""
import json
import sys

import os

tag = sys.argv[1] if len(sys.argv) > 1 else ""
files = sys.argv[2:]
if not files and os.path.isfile(tag):              # "summarize.py file.json": the tag was left out
    files, tag = [tag], os.path.basename(tag)
if not files and sys.stdin.isatty():
    sys.exit("usage: bench.py ... | summarize.py [tag]   or   summarize.py tag file.json ...")


def lines():
    if files:
        for f in files:
            text = open(f).read()
            try:                                   # a pretty-printed JSON document (profiles/*.json)
                yield json.dumps(json.loads(text))
            except ValueError:
                for l in text.splitlines():
                    yield l
    else:
        for l in sys.stdin:
            yield l


for l in lines():
    if not l.startswith("{"):
        print(l[:300].rstrip())
        continue
    d = json.loads(l)
    print("[%s] x real time %.0f  ms/step %.1f  stages %s" % (tag, d["value"], d["ms_per_step"],
          {k: round(v, 2) for k, v in d.get("stage_ms", {}).items()}))
    if "decoder" in d:
        print("    decoder %s" % {k: round(v, 1) for k, v in d["decoder"].items()})
    for key in ("roofline", "roofline_other_stage"):
        r = d.get(key)
        if r:
            print("    %s: %s %.1f %s = %.3f of peak" % (key, r.get("kernel", ""), r["achieved"], r["unit"], r["frac"]))
    c = d.get("cpu_baseline")
    if c:
        print("    cpu_baseline %.1f %s on %d threads (%s)" % (c["value"], c["unit"], c["cores"], c["kind"]))
