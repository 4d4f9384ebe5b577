# rocprofv3 kernel stats + PMC passes on the headline bench (round 2).  Run from the repo root on the GPU box.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --no-cpu-baseline --no-wer --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_stats -o stats -- $B > $R/gpurun_out/r02_prof_stats.json 2> $R/gpurun_out/r02_prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r02_prof_fetch -o fetch -- $B > $R/gpurun_out/r02_prof_fetch.json 2> $R/gpurun_out/r02_prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r02_prof_write -o write -- $B > $R/gpurun_out/r02_prof_write.json 2> $R/gpurun_out/r02_prof_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $R/gpurun_out/r02_prof_sq -o sq -- $B > $R/gpurun_out/r02_prof_sq.json 2> $R/gpurun_out/r02_prof_sq.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/r02_prof_tcc -o tcc -- $B > $R/gpurun_out/r02_prof_tcc.json 2> $R/gpurun_out/r02_prof_tcc.err
ls -la $R/gpurun_out/r02_prof_*/ | head -40
find $R/gpurun_out/r02_prof_stats -name "*.csv" | head
