# per-launch GEMM durations at batch 512 (one forward): which layers are far from the fp32 MFMA peak
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gemm
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm -o run -- python3 bench.py --utts 512 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/gemm/bench.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gemm/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "TdnnGemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 2          # two forwards (unsliced pass + timed step): take the last one
for r in rows[-n:]:
    print(r["Kernel_Name"].split("(")[0][-28:], "grid", r["Grid_Size_X"], r["Grid_Size_Y"], "wg", r["Workgroup_Size_X"], "us %.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
