"""The search kernels sit at the edge of the register file (1024 threads per CU: 128 VGPRs per thread).  hipcc's
allocator has two regimes there: the one the kernels are written for (AdvanceKernel without scratch, DecodeQueueKernel
with ~130 B of it per lane: loop-invariant pointers), and one in which it spills several hundred bytes per lane -- builds
in that second regime have produced WRONG lattices on the device (round 3: three unrelated source changes tipped it, each
time `tests/test_gpu_decoder.py` failed until the change was withdrawn).  This test compiles decoder.hip with
-Rpass-analysis=kernel-resource-usage and fails when a change has tipped the allocator, before anything runs on a GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_search_kernels_stay_in_the_register_regime_they_were_written_for(tmp_path):
    src = os.path.join(ROOT, "kaldi_amd", "csrc", "decoder.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-DKAMD_NT=1024", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-Wno-unused-result", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-c", "-o",
                        str(tmp_path / "decoder.o"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                usage[name][key] = int(m.group(1))
    adv = [v for k, v in usage.items() if "AdvanceKernel" in k]
    dq = [v for k, v in usage.items() if "DecodeQueueKernel" in k]
    assert adv and dq, sorted(usage)
    assert adv[0]["VGPRs"] <= 128 and dq[0]["VGPRs"] <= 128          # 16 wavefronts per CU
    assert adv[0]["ScratchSize [bytes/lane]"] == 0, usage
    assert dq[0]["ScratchSize [bytes/lane]"] <= 256, usage
    shutil.rmtree(tmp_path, ignore_errors=True)
