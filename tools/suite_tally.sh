#!/bin/bash
# N consecutive runs of the whole GPU suite on one box ("next round" 2(d)): one line per run in $O/tally.txt,
# the full log of any run that does not pass kept beside it.   usage: tools/suite_tally.sh <runs> <out dir>
N=${1:-10}; O=${2:-gpurun_out/tally}; mkdir -p $O
for i in $(seq 1 $N); do
  python -m pytest tests -q -m gpu -p no:cacheprovider > $O/run_$i.txt 2>&1
  rc=$?
  echo "run $i rc $rc: $(tail -1 $O/run_$i.txt)" >> $O/tally.txt
  if [ $rc -eq 0 ]; then rm -f $O/run_$i.txt; fi
done
cat $O/tally.txt
