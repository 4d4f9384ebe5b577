#!/bin/bash
# Runs on the GPU box (gpurun): default bench line, rocprofv3 kernel stats of the same command,
# separate PMC passes (FETCH_SIZE / WRITE_SIZE), and the load / batch sweep.  Everything lands
# in gpurun_out/round/; tools/collect_profiles.py turns it into profiles/rNN_*.
# usage: tools/profile_round.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/round
rm -rf $O; mkdir -p $O
timeout 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/pmc_mfma -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq1.json 2> $O/pmc_sq1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq2 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq2.json 2> $O/pmc_sq2.err
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq3.json 2> $O/pmc_sq3.err
for cfg in "light --ll-std 2.0" "saturated --ll-std 1.0" "b256 --utts 256" "b512 --utts 512"; do
  set -- $cfg; name=$1; shift
  timeout 400 python3 bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
done
# the i-vector variant (secondary measurement): bench line + kernel stats
timeout 400 python3 bench.py --ivectors --ll-std 1.8 --no-cpu-baseline > $O/bench_ivectors.json 2> $O/bench_ivectors.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iv -o run -- python3 bench.py --ivectors --ll-std 1.8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ivectors_under_rocprof.json 2> $O/stats_iv.err
# keep the merged directory small: only the per-kernel CSVs
find $O -name "*_kernel_trace.csv" -size +20M -delete
ls -la $O $O/*/* 2>/dev/null | head -40
