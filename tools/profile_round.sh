#!/bin/bash
# One round's profiles, on the GPU box (gpurun): the default bench line (BASELINE configs[2]: the recipe-faithful headline and its legs),
# then -- on the headline alone (--no-random-leg, no CPU / WER / streaming legs) -- rocprofv3 kernel stats and separate PMC
# passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil / SQ / TCC), the kernel stats of round 3's headline (--headline random) and of
# the streaming passes.  Everything lands in gpurun_out/round5/; tools/collect_profiles5.py turns it into profiles/r05_*.
# usage: tools/profile_round.sh rNN [quick]      (quick: kernel stats + FETCH / WRITE / SQ passes only)
set -u
export TMPDIR=/tmp
case "${1:-}" in r[0-9][0-9]) ;; *) echo "usage: $0 rNN [quick]" >&2; exit 2;; esac
O=gpurun_out/round$((10#${1#r}))
shift
rm -rf $O; mkdir -p $O
H="--no-random-leg --no-planted --no-ivector-leg --no-cpu-baseline --no-wer --no-streaming"
B="python3 bench.py --steps 2 --warmup 1 $H"
timeout 1200 python3 bench.py --verbose > $O/bench_default.json 2> $O/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.json 2> $O/pmc_write.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq1 -o run -- $B > $O/pmc_sq1.json 2> $O/pmc_sq1.err
if [ "${1:-}" != "quick" ]; then
  timeout 400 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/pmc_mfma -o run -- $B > $O/pmc_mfma.json 2> $O/pmc_mfma.err
  timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -o run -- $B > $O/pmc_sq3.json 2> $O/pmc_sq3.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_random -o run -- python3 bench.py --steps 2 --warmup 1 --headline random --no-bracket --no-planted --no-ivector-leg --no-cpu-baseline --no-wer --no-streaming > $O/bench_random_under_rocprof.json 2> $O/stats_random.err
fi
find $O -name "*_kernel_trace.csv" -size +20M -delete
ls -la $O | head -30
