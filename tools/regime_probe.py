#!/usr/bin/env python3
"""Register-regime probe for the search kernels.

Round 3 saw builds of decoder.hip that the register allocator had pushed into spilling decode WRONGLY.  This tool
builds the library several times with the SAME sources and different code-generation switches that force each kind
of spill on purpose, and runs the bit-exact decoder tests against every build, so that "which kind of spill breaks
the result" is an experiment and not a guess:

  base     the product flags (control)
  mlicm    machine LICM left on (round 3's flags)
  s2m      -mllvm -amdgpu-spill-sgpr-to-vgpr=0: every SGPR spill goes to scratch MEMORY instead of VGPR lanes
  v64      search kernels compiled for a 64-VGPR budget: forced VGPR spills (SGPR spills stay in lanes)
  s2m_v64  both: the closest synthetic stand-in for round 3's bad regime (VGPR spills + SGPR spills to memory)
  mlicm_s2m_v96  the same with machine LICM on

    python tools/regime_probe.py build            # here (no GPU): build/regime/libkaldi_amd_<variant>.so + resources.json
    python tools/regime_probe.py run [tests...]   # on the GPU box: pytest per variant -> gpurun_out/regime/<variant>.log

Extra variants: KAMD_REGIME_EXTRA="name:flag flag;name2:flag" (flags appended to the decoder.hip compile only).
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kaldi_amd", "csrc")
OUT = os.path.join(ROOT, "build", "regime")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["--offload-arch=gfx950", "-DKAMD_NT=1024", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
        "-Wno-unused-result", "-Wno-pass-failed"]
NO_MLICM = ["-mllvm", "-disable-machine-licm"]          # kaldi_amd/csrc/Makefile: DECODER_FLAGS
S2M = ["-mllvm", "-amdgpu-spill-sgpr-to-vgpr=0"]
VARIANTS = {
    "base": NO_MLICM,                                        # the product flags (control)
    "mlicm": [],                                             # machine LICM left on (round 3's flags)
    "s2m": NO_MLICM + S2M,                                   # every SGPR spill to scratch memory
    "v64": NO_MLICM + ["-DKAMD_SEARCH_VGPRS=64"],            # half the VGPR budget: forced VGPR spills
    "s2m_v64": NO_MLICM + S2M + ["-DKAMD_SEARCH_VGPRS=64"],  # both
    "mlicm_s2m_v96": S2M + ["-DKAMD_SEARCH_VGPRS=96"],       # round 3's flags, spilling both ways
}
for item in filter(None, os.environ.get("KAMD_REGIME_EXTRA", "").split(";")):
    name, flags = item.split(":", 1)
    VARIANTS[name] = flags.split()
PER_VARIANT_S = int(os.environ.get("KAMD_REGIME_TIMEOUT", "420"))
DEFAULT_TESTS = ["tests/test_gpu_decoder.py", "tests/test_gpu_queue.py", "tests/test_gpu_search_mode.py"]


def resources(stderr):
    usage, name = {}, None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    return {k: v for k, v in usage.items() if "AdvanceKernel" in k or "DecodeQueueKernel" in k}


def build():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    objs = [os.path.join(ROOT, "kaldi_amd", "build", f) for f in sorted(os.listdir(os.path.join(ROOT, "kaldi_amd", "build")))
            if f.endswith(".o") and f != "decoder.o"]
    report = {}
    for name, extra in VARIANTS.items():
        obj = os.path.join(OUT, "decoder_%s.o" % name)
        r = subprocess.run([HIPCC] + BASE + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", obj,
                                                      os.path.join(CSRC, "decoder.hip")], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr[-3000:])
        lib = os.path.join(OUT, "libkaldi_amd_%s.so" % name)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [obj])
        os.remove(obj)
        report[name] = {"flags": extra, "kernels": resources(r.stderr)}
        for k, v in report[name]["kernels"].items():
            print("%-8s %-22s VGPRs %3d  scratch %4d B/lane  SGPR spills %3d  VGPR spills %3d" % (
                name, "Advance" if "Advance" in k else "DecodeQueue", v.get("VGPRs", -1), v.get("ScratchSize [bytes/lane]", -1),
                v.get("SGPRs Spill", -1), v.get("VGPRs Spill", -1)))
    with open(os.path.join(OUT, "resources.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


def run(tests):
    out = os.path.join(ROOT, "gpurun_out", "regime")
    os.makedirs(out, exist_ok=True)
    summary = {}
    for name in VARIANTS:
        lib = os.path.join(OUT, "libkaldi_amd_%s.so" % name)
        if not os.path.exists(lib):
            continue
        env = dict(os.environ, KAMD_LIB=lib)
        # a build that decodes wrongly may also spin (a fixpoint loop fed garbage): every test and the whole run are bounded,
        # so that a variant can fail without taking the GPU box with it
        try:
            r = subprocess.run(["timeout", "-k", "10", str(PER_VARIANT_S), sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-rf", "--tb=line",
                                "-p", "no:cacheprovider", "--timeout=90", "--timeout-method=thread"] + tests,
                               cwd=ROOT, env=env, capture_output=True, text=True)
        except Exception as e:                      # noqa: BLE001
            summary[name] = {"rc": -1, "tail": repr(e), "failed": []}
            continue
        with open(os.path.join(out, name + ".log"), "w") as f:
            f.write(r.stdout[-20000:] + "\n" + r.stderr[-4000:])
        tail = [l for l in r.stdout.splitlines() if l.strip()][-1:] or ["?"]
        summary[name] = {"rc": r.returncode, "tail": tail[0],
                         "failed": [l.split(" ")[1] for l in r.stdout.splitlines() if l.startswith("FAILED ")][:40]}
        print(name, r.returncode, tail[0], flush=True)
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == "run":
        run(sys.argv[2:] or DEFAULT_TESTS)
    else:
        sys.exit(__doc__)
