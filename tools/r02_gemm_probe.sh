export TMPDIR=/tmp
for G in 1 0; do echo "== KAMD_GEMM_GEN1=$G"; KAMD_GEMM_GEN1=$G python3 tools/gemm_probe.py 400000 2>&1 | grep -v amdgpu; done
for G in 1 0; do
rm -rf gpurun_out/gp_$G; KAMD_GEMM_GEN1=$G rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/gp_$G -o run -- python3 tools/gemm_probe.py 400000 > /dev/null 2> gpurun_out/gp_$G.err
python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("gpurun_out/gp_$G/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "TdnnGemm" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0].replace("void kamd::","")
            acc.setdefault(k,{}).setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
for k,v in acc.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    wc=m.get("SQ_WAVE_CYCLES",1)
    print("gen1=$G", k, {c:round(x/wc,3) for c,x in m.items() if c!="SQ_WAVE_CYCLES"}, "busy_cycles", m.get("SQ_BUSY_CYCLES"))
PY
done
