"""Invariants of the search kernels' source that no GPU test can see in minutes.

A step soak found a wild lookup in HCLG's state-offset array once in a few hundred bench steps; since then every lookup of
that array in device code sits behind a range check of the state and stops the lane with ERR_BAD_STATE(site).  This test
keeps it that way: a new `d.g.off[...]` without the check fails here, on the CPU, and the bits a lane can report are all
named in bench.py's failure report."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "kaldi_amd", "csrc", "decoder.hip")


def _lines():
    with open(SRC) as f:
        return f.read().splitlines()


def test_every_graph_offset_lookup_is_range_checked():
    lines = _lines()
    sites = [i for i, l in enumerate(lines) if "d.g.off[" in l]
    assert len(sites) >= 6, sites
    for i in sites:
        window = "\n".join(lines[max(0, i - 16):i + 1])
        assert "d.g.num_states" in window and "ERR_BAD_STATE(" in window, \
            "decoder.hip:%d indexes the state-offset array without a range check in front of it:\n%s" % (i + 1, lines[i])


def test_bad_state_sites_are_distinct_and_named_in_the_bench_report():
    src = "\n".join(_lines())
    sites = sorted(int(m) for m in re.findall(r"ERR_BAD_STATE\((\d+)\)", src))
    assert sites == sorted(set(sites)) and sites[0] >= 1, sites            # one bit per lookup
    m = re.search(r"#define ERR_BAD_STATE\(site\) \(ERR_INTERNAL \| \((\d+) << \(site\)\)\)", src)
    assert m, "ERR_BAD_STATE changed its form"
    base = int(m.group(1))
    import bench
    named = {bit for bit, _ in bench.ERROR_FLAGS}
    for s in sites:
        assert (base << s) in named, "bench.ERROR_FLAGS does not name bit %d (ERR_BAD_STATE(%d))" % (base << s, s)
    enum = re.search(r"enum \{ ERR_HASH = 1, ERR_TOK = 2, ERR_LINK = 4, ERR_FRAMES = 8, ERR_WL = 16, ERR_INTERNAL = 32 \};", src)
    assert enum and {1, 2, 4, 8, 16, 32, 64} <= named


def test_second_chance_covers_lanes_stopped_by_an_internal_check():
    """Arena / pool exhaustion (bits 2, 4, 64), a frame's level-2 share filled up (1: the wide launch addresses the whole
    table) and a lane stopped by an internal check (32, with or without a bad-state bit) take the second chance -- the
    search is deterministic and the event transient; a frame overflow (8) does not.  Internal-check stops are counted on
    their own (n_internal_events), whether or not the second chance is on."""
    with open(os.path.join(ROOT, "kaldi_amd", "csrc", "batch.cc")) as f:
        b = f.read()
    assert "(o.rec.error & (1 | 2 | 4 | 32 | 64)) != 0 && (o.rec.error & 8) == 0" in b
    assert b.index("n_internal++") < b.index("const bool retry_on")


def test_soak_variant_transforms_still_match_the_source():
    """tools/soak_variants.py builds the experiment libraries of the step soak from textual transforms of decoder.hip: every
    transform must still find its text (a silent no-op would make a soak compare a build with itself)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import soak_variants as sv
    src = "\n".join(_lines()) + "\n"
    for name in ("noguards", "guards_commit", "guards_rest", "diag", "v110", "sc1"):
        out = sv.transform(name, src)
        assert out != src, name
    assert sv.transform("noguards", src).count("if (false)") == 6
    assert "-disable-machine-licm" in sv.product_flags() and "--offload-arch=gfx950" in sv.product_flags()


def test_every_copy_of_the_library_goes_through_the_bounce_buffer_wrappers():
    """common.h turns hipMemcpy / hipMemcpyAsync / hipMemcpy2D / hipMemcpy2DAsync into kamd::Memcpy*Safe (pageable host memory
    through the library's page-locked bounce buffers: DESIGN.md section 0).  Every source that copies must therefore see
    common.h, no source may use a copy call of the runtime that has no wrapper, and in common.cc -- the one file that sees
    the runtime's own functions -- every raw call is inside the wrappers: the EXPORTED copies (kamd_memcpy_h2d / _d2h) go
    through MemcpySafe like everybody else."""
    import re
    csrc = os.path.join(ROOT, "kaldi_amd", "csrc")
    wrapped = {"hipMemcpy", "hipMemcpyAsync", "hipMemcpy2D", "hipMemcpy2DAsync"}
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cc", ".hip", ".h")):
            continue
        text = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", text)
        called = set(re.findall(r"\b(hipMemcpy\w*)\s*\(", code))
        called = {c for c in called if not c.startswith("hipMemcpyKind")}
        if f == "common.cc":
            assert "#define KAMD_RAW_MEMCPY" in text
            # raw calls are written ::hipMemcpy...( and occur only before the first extern "C" (i.e. inside the wrappers)
            head, _, exports = code.partition('extern "C"')
            assert exports and "hipMemcpy" not in re.sub(r"hipMemcpy(HostToDevice|DeviceToHost)", "", exports), "an exported function of common.cc copies raw"
            assert re.search(r"int kamd_memcpy_h2d\([^)]*\) \{ KAMD_HIP\(kamd::MemcpySafe\(", exports)
            assert re.search(r"int kamd_memcpy_d2h\([^)]*\) \{ KAMD_HIP\(kamd::MemcpySafe\(", exports)
            assert all(("::" + c + "(") in head for c in wrapped)
            assert called <= wrapped, called
            continue
        if f == "common.h":
            for c in wrapped:
                assert re.search(r"#define %s\(\.\.\.\) kamd::Memcpy\w*Safe\(__VA_ARGS__\)" % c, text), c
            continue
        assert "::hipMemcpy" not in text and "KAMD_RAW_MEMCPY" not in text, f
        assert called <= wrapped, (f, called - wrapped)          # (no hipMemcpyDtoH / HtoD / WithStream / 3D / Peer ... without a wrapper)
        if called:
            assert '#include "common.h"' in text, f
    # the caller's samples are page-locked in place only over the pages that lie wholly inside them (batch.cc, load_host)
    b = open(os.path.join(csrc, "batch.cc")).read()
    assert b.count("hipHostRegister(") == 1 and "hipHostRegister(reinterpret_cast<void *>(in_lo), in_hi - in_lo" in b
    # what counts as page-locked is the library's own list, never the runtime's (stale) view: no hipPointerGetAttributes in the
    # wrappers, and every hipHostMalloc / hipHostFree / hipHostRegister / hipHostUnregister of the library is noted there
    cc = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, "common.cc")).read())
    assert "hipPointerGetAttributes" not in cc and "g_pinned" in cc
    h = open(os.path.join(csrc, "common.h")).read()
    for name, to in (("hipHostMalloc(...)", "HostMallocNotedT(__VA_ARGS__)"), ("hipHostFree(p)", "HostFreeNoted(p)"),
                     ("hipHostRegister(...)", "HostRegisterNoted(__VA_ARGS__)"), ("hipHostUnregister(p)", "HostUnregisterNoted(p)")):
        assert "#define %s kamd::%s" % (name, to) in h, name
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cc", ".hip")) and f != "common.cc":
            text = open(os.path.join(csrc, f)).read()
            assert "::hipHost" not in text and "hipPointerGetAttributes" not in text, f
            if "hipHostMalloc" in text or "hipHostRegister" in text:
                assert '#include "common.h"' in text or '#include "meta_ring.h"' in text, f
    # the hpp mirror and the examples are host programs over the C-ABI: they copy nothing themselves
    for f in (os.path.join(ROOT, "include", "kaldi_amd.hpp"),) + tuple(os.path.join(ROOT, "examples", x) for x in os.listdir(os.path.join(ROOT, "examples"))):
        assert "hipMemcpy" not in open(f).read(), f


def test_the_exported_copies_do_not_call_the_runtime_directly():
    """The shipped library, disassembled: kamd_memcpy_h2d / kamd_memcpy_d2h call kamd::MemcpySafe, not hipMemcpy@plt."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "kaldi_amd", "lib", "libkaldi_amd.so")
    if not os.path.exists(lib) or not shutil.which("objdump"):
        import pytest
        pytest.skip("no built library / objdump")
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", lib], capture_output=True, text=True, check=True).stdout
    for sym in ("kamd_memcpy_h2d", "kamd_memcpy_d2h"):
        body = dis[dis.index("<%s>:" % sym):]
        body = body[:body.index("\n\n")]
        calls = [ln for ln in body.splitlines() if "call" in ln]
        assert any("MemcpySafe" in c for c in calls), (sym, calls)
        assert not any("<hipMemcpy" in c for c in calls), (sym, calls)


def test_a_dying_gpu_run_leaves_its_last_words_in_the_log():
    """pytest.ini captures at the sys level (the HSA runtime's fault line and glibc's messages go to fd 2, which fd-level capture
    swallows when the process dies) and conftest switches the library's SIGABRT backtrace on."""
    ini = open(os.path.join(ROOT, "pytest.ini")).read()
    assert "--capture=sys" in ini
    conf = open(os.path.join(ROOT, "tests", "conftest.py")).read()
    assert 'os.environ.setdefault("KAMD_ABORT_BACKTRACE", "1")' in conf
    assert "KAMD_ABORT_BACKTRACE" in open(os.path.join(ROOT, "kaldi_amd", "csrc", "common.cc")).read()
