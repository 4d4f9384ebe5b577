#!/bin/bash
# usage: tools/sweep.sh "<bench args A>" "<bench args B>" ...   (prints a compact summary per run)
for a in "$@"; do
  timeout 400 python bench.py $a 2>/tmp/bench_err.log | python tools/summarize.py "$a" || tail -5 /tmp/bench_err.log
done
