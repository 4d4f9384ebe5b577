"""The search kernels run as 1024-thread lanes, one per CU: 128 VGPRs per thread at most.  Round 3's builds sat at that
edge (AdvanceKernel 121, DecodeQueueKernel 128 + 31 spilled) and source edits that tipped the allocator into spilling to
scratch MEMORY decoded wrongly on the device.  Round 4 took the kernels off the edge: the launch
descriptors are re-read per phase instead of living across the whole kernel, the thread id is opaque to loop-invariant
code motion, and decoder.hip is compiled without MachineLICM.  This test holds the line before anything runs on a GPU:
compiled with the product's flags for a budget of 120 VGPRs (KAMD_SEARCH_VGPRS), both kernels must stay out of scratch
memory altogether -- i.e. they carry at least 8 VGPRs of head-room against the hardware limit -- and spill no VGPR."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def product_flags():
    """FLAGS + DECODER_FLAGS of kaldi_amd/csrc/Makefile (read from it, so that the test compiles what the product does)."""
    mk = open(os.path.join(ROOT, "kaldi_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^FLAGS = (.*)$", mk, re.M).group(1)
    dec = re.search(r"^DECODER_FLAGS = (.*)$", mk, re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").replace("$(KAMD_NT)", "1024")
    return flags.split() + dec.split()


def kernel_usage(extra=()):
    src = os.path.join(ROOT, "kaldi_amd", "csrc", "decoder.hip")
    tmp = os.path.join(ROOT, "kaldi_amd", "build", "_resources")
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([HIPCC] + product_flags() + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o",
                                                                   os.path.join(tmp, "decoder.o"), src], capture_output=True, text=True)
    shutil.rmtree(tmp, ignore_errors=True)
    assert r.returncode == 0, r.stderr[-2000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    return usage


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_search_kernels_have_register_head_room_and_no_scratch():
    usage = kernel_usage()
    adv = [v for k, v in usage.items() if "AdvanceKernel" in k]
    dq = [v for k, v in usage.items() if "DecodeQueueKernel" in k]
    assert adv and dq, sorted(usage)
    for k in (adv[0], dq[0]):
        assert k["VGPRs"] <= 120, usage                      # >= 8 registers below the 128 a 1024-thread workgroup may use
        assert k["ScratchSize [bytes/lane]"] == 0, usage     # nothing in scratch memory: no VGPR spill, no SGPR spill to memory
        assert k["VGPRs Spill"] == 0, usage
    # every other 1024-thread kernel of the file too: none may need scratch
    for name, k in usage.items():
        assert k.get("ScratchSize [bytes/lane]", 0) == 0, (name, k)
        assert k.get("VGPRs", 0) <= 128, (name, k)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_ivector_solver_fits_two_workgroups_per_cu():
    """SolveKernel (one sequential CG chain per utterance) is bound by how many chains a CU holds at once.  Rounds 2-5's build
    took 256 VGPRs + 160 AGPRs per thread -- one 256-thread workgroup per CU, 256 chains on the device -- because the next
    step's increment was prefetched into 66 registers and 64 packed-triangle positions were kept alive across the steps.
    Round 6 (DMA prefetch into LDS, positions recomputed per step): at most 256 registers, nothing in scratch, two workgroups
    per CU."""
    mk = open(os.path.join(ROOT, "kaldi_amd", "csrc", "Makefile")).read()
    flags = re.search(r"^FLAGS = (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(KAMD_NT)", "1024").split()
    tmp = os.path.join(ROOT, "kaldi_amd", "build", "_resources_iv")
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([HIPCC] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.path.join(tmp, "ivector.o"),
                                          os.path.join(ROOT, "kaldi_amd", "csrc", "ivector.hip")], capture_output=True, text=True)
    shutil.rmtree(tmp, ignore_errors=True)
    assert r.returncode == 0, r.stderr[-2000:]
    block = r.stderr[r.stderr.index("SolveKernel"):]
    get = lambda what: int(re.search(r"remark:\s+%s: (\d+)" % re.escape(what), block).group(1))      # noqa: E731
    assert get("VGPRs") + get("AGPRs") <= 256 and get("ScratchSize [bytes/lane]") == 0 and get("VGPRs Spill") == 0
    assert get("Occupancy [waves/SIMD]") >= 2
