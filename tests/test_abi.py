"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/kaldi_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from kaldi_amd import _lib, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    h = open(os.path.join(ROOT, "include", "kaldi_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(kamd_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = C.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    # and the python binding table covers the same set
    assert sorted(_lib.EXPORTS) == syms


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(abi.Arc) == 16                      # == fst::StdArc
    assert C.sizeof(abi.FrameOpts) == 40
    assert C.sizeof(abi.MelOpts) == 24
    assert C.sizeof(abi.MfccOpts) == 40 + 24 + 24
    assert C.sizeof(abi.DecoderConfig) == 32
    assert C.sizeof(abi.DecodeTask) == 24
    assert abi.LAT_ARC_DTYPE.itemsize == 24 and abi.ARC_DTYPE.itemsize == 16


def test_defaults_match_reference_option_structs():
    L = _lib.lib()
    c = abi.DecoderConfig()
    L.kamd_decoder_config_default(C.byref(c))
    # decoder/lattice-faster-decoder.h:56-64
    assert (c.beam, c.max_active, c.min_active, c.lattice_beam) == (16.0, 2147483647, 200, 10.0)
    assert (c.prune_interval, c.beam_delta, c.hash_ratio) == (25, 0.5, 2.0)
    assert abs(c.prune_scale - 0.1) < 1e-7
    m = abi.MfccOpts()
    L.kamd_mfcc_opts_default(C.byref(m))
    # feat/feature-mfcc.h:50-58, feat/feature-window.h:54-66 (dither forced to 0)
    assert (m.mel.num_bins, m.num_ceps, m.use_energy, m.cepstral_lifter) == (23, 13, 1, 22.0)
    assert (m.frame.samp_freq, m.frame.frame_shift_ms, m.frame.frame_length_ms) == (16000.0, 10.0, 25.0)
    assert m.frame.dither == 0.0 and abs(m.frame.preemph_coeff - 0.97) < 1e-7


def test_product_fails_loudly_without_gpu():
    """No silent CPU fallback: without a device the product raises."""
    L = _lib.lib()
    if L.kamd_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.KamdError):
        _lib.require_gpu()
    from kaldi_amd import feat
    with pytest.raises(_lib.KamdError):
        feat.Mfcc()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under kaldi_amd/ may import, link, dlopen or
    call it (comments may mention it)."""
    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|oracle/|\borc_[a-z]|orc\.py")
    for top in ("kaldi_amd", "tools", "examples", "include"):          # the product, its command lines and its headers
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith((".py", ".hip", ".cc", ".h", ".hpp", ".sh", "Makefile")):
                    src = open(os.path.join(dp, f)).read()
                    assert not pat.search(src), os.path.join(dp, f)
