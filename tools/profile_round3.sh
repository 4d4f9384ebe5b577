#!/bin/bash
# Round 3, on the GPU box (gpurun): the default bench line (BASELINE configs[2], all legs), then -- on the headline load
# only (--no-bracket --no-planted --no-ivector-leg) -- rocprofv3 kernel stats and separate PMC passes (FETCH_SIZE /
# WRITE_SIZE / MfmaUtil / SQ / TCC), and the kernel stats of the --ivectors variant.  Everything lands in
# gpurun_out/round3/; tools/collect_profiles3.py turns it into profiles/r03_*.   usage: tools/profile_round3.sh
set -u
export TMPDIR=/tmp
O=gpurun_out/round3
rm -rf $O; mkdir -p $O
H="--no-bracket --no-planted --no-ivector-leg --no-cpu-baseline --no-wer"
B="python3 bench.py --steps 2 --warmup 1 $H"
timeout 900 python3 bench.py --verbose > $O/bench_default.json 2> $O/bench_default.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iv -o run -- $B --ivectors > $O/bench_ivectors_under_rocprof.json 2> $O/stats_iv.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.json 2> $O/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/pmc_mfma -o run -- $B > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq1 -o run -- $B > $O/pmc_sq1.json 2> $O/pmc_sq1.err
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq3 -o run -- $B > $O/pmc_sq3.json 2> $O/pmc_sq3.err
find $O -name "*_kernel_trace.csv" -size +20M -delete
ls -la $O | head -30
