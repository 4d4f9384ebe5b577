"""On-disk formats: OpenFst binary HCLG files and Kaldi lattice archives.
PARITY UNPINNED: the reference tree holds no OpenFst file (its two .fst fixtures are empty),
so these are write -> read round trips plus byte-level checks of the facts the reference does
state (magic byte 214, type strings, the text form of lat/kaldi-lattice.cc)."""
import struct

import numpy as np
import pytest

from kaldi_amd import io as kio
from kaldi_amd import abi, synth
from kaldi_amd._lib import KamdError
from oracle import orc


def same_fst(a, b):
    return (a.num_states == b.num_states and a.start == b.start
            and np.array_equal(a.arc_off, b.arc_off) and np.array_equal(a.arcs, b.arcs)
            and np.array_equal(a.final.view(np.uint32), np.asarray(b.final, np.float32).view(np.uint32)))


@pytest.mark.parametrize("fst_type,align", [("vector", False), ("const", False), ("const", True)])
def test_hclg_round_trip(tmp_path, fst_type, align):
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=1)
    p = tmp_path / "HCLG.fst"
    kio.write_openfst(p, g, fst_type, align)
    raw = p.read_bytes()
    assert raw[0] == 214                                   # lat/kaldi-lattice.cc:377
    assert struct.unpack("<i", raw[:4])[0] == 2125659606
    n = struct.unpack("<i", raw[4:8])[0]
    assert raw[8:8 + n].decode() == fst_type
    m = struct.unpack("<i", raw[8 + n:12 + n])[0]
    assert raw[12 + n:12 + n + m].decode() == "standard"
    assert same_fst(kio.read_openfst(p), g)


def test_const_layout_sizes(tmp_path):
    """ConstFst: 20-byte state records and 16-byte arcs after the header."""
    g = synth.make_random_graph(num_states=50, seed=3)
    p = tmp_path / "g.fst"
    kio.write_openfst(p, g, "const", False)
    header = 4 + (4 + 5) + (4 + 8) + 4 + 4 + 8 + 8 + 8 + 8
    assert p.stat().st_size == header + 20 * g.num_states + 16 * g.num_arcs
    kio.write_openfst(p, g, "const", True)
    pad = lambda x: (x + 15) // 16 * 16
    assert p.stat().st_size == pad(pad(header) + 20 * g.num_states) + 16 * g.num_arcs


def test_symbol_tables_are_skipped(tmp_path):
    g = synth.make_random_graph(num_states=12, seed=5)
    p = tmp_path / "v.fst"
    kio.write_openfst(p, g, "vector")
    raw = bytearray(p.read_bytes())
    # splice an input symbol table in after the header and set HAS_ISYMBOLS
    hdr = 4 + (4 + 6) + (4 + 8) + 4 + 4 + 8 + 8 + 8 + 8
    flags_at = 4 + (4 + 6) + (4 + 8) + 4
    raw[flags_at:flags_at + 4] = struct.pack("<i", 1)
    def s(x):
        return struct.pack("<i", len(x)) + x
    sym = struct.pack("<i", 2125658996) + s(b"words") + struct.pack("<qq", 2, 2)
    sym += s(b"<eps>") + struct.pack("<q", 0) + s(b"hello") + struct.pack("<q", 1)
    p.write_bytes(bytes(raw[:hdr]) + sym + bytes(raw[hdr:]))
    assert same_fst(kio.read_openfst(p), g)


def test_bad_files_are_rejected(tmp_path):
    p = tmp_path / "x.fst"
    p.write_bytes(b"not an fst")
    with pytest.raises(KamdError, match="OpenFst"):
        kio.read_openfst(p)
    g = synth.make_random_graph(num_states=12, seed=5)
    kio.write_openfst(p, g, "const")
    raw = p.read_bytes()
    p.write_bytes(raw[:-7])
    with pytest.raises(KamdError, match="truncated"):
        kio.read_openfst(p)
    with pytest.raises(KamdError, match="cannot open"):
        kio.read_openfst(tmp_path / "missing.fst")


def _decode_lattice():
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=2)
    ll, _, _ = synth.sample_utterance(g, n_words=4, seed=3, peak=5.0)
    d = orc.Decoder(g, abi.decoder_config_recipe(), 1)
    d.Decode(ll)
    return d.GetRawLattice()


@pytest.mark.parametrize("binary", [True, False])
def test_lattice_archive_round_trip(tmp_path, binary):
    lat = _decode_lattice()
    p = tmp_path / "lat.1"
    kio.write_lattice(p, "utt-a", lat, binary=binary, append=False)
    kio.write_lattice(p, "utt-b", lat, binary=binary, append=True, acoustic_scale=0.5)
    got = list(kio.read_lattices(p))
    assert [k for k, _, _, _ in got] == ["utt-a", "utt-b"]
    start, fin, arcs = kio.lattice_arrays(lat)
    for (key, st, f, a), scale in zip(got, (1.0, 0.5)):
        assert st == start and a.size == arcs.size
        for fld in ("src", "dst", "ilabel", "olabel"):
            assert np.array_equal(a[fld], arcs[fld])
        tol = 0 if binary else 1e-5           # text form prints 6 significant digits
        assert np.allclose(a["graph_cost"], arcs["graph_cost"], rtol=tol, atol=0)
        assert np.allclose(a["acoustic_cost"], arcs["acoustic_cost"] / np.float32(scale), rtol=tol, atol=0)
        assert np.array_equal(np.isfinite(f), np.isfinite(fin))
        assert np.allclose(f[np.isfinite(f)], fin[np.isfinite(fin)], rtol=tol)


def test_lattice_text_form(tmp_path):
    """key, newline, FstPrinter lines with the start state first, empty line
    (lat/kaldi-lattice.cc:96-130; weights 'graph,acoustic', fstext/lattice-weight.h:396-404)."""
    arcs = np.zeros(3, abi.LAT_ARC_DTYPE)
    arcs[0] = (0, 2, 0, 7, 0.0, 0.0)            # One() weight is omitted
    arcs[1] = (1, 0, 5, 0, 1.5, -2.25)
    arcs[2] = (2, 2, 3, 3, 0.5, np.inf)
    fin = np.array([np.inf, np.inf, np.inf, np.inf, 0.25, 0.0], np.float32)

    class L:
        pass
    lat = L()
    lat.frame = np.zeros(3, np.int32); lat.start = 1; lat.arcs = arcs
    lat.final = np.array([np.inf, np.inf, 0.25], np.float32)
    p = tmp_path / "t.lat"
    kio.write_lattice(p, "k1", lat, binary=False, append=False)
    assert p.read_text() == "k1 \n1\t0\t5\t0\t1.5,-2.25\n0\t2\t0\t7\n2\t2\t3\t3\t0.5,Infinity\n2\t0.25,0\n\n"
    (key, st, f, a), = list(kio.read_lattices(p))
    assert key == "k1" and st == 1 and a.size == 3 and f[4] == np.float32(0.25) and f[5] == 0.0


def test_reader_refuses_header_fields_it_was_not_written_for(tmp_path):
    """The OpenFst layout is restated from the published format and has never met a file OpenFst itself wrote: unknown
    flag bits, an unknown file version or a wrong magic number are refused BY NAME instead of being read on a guess."""
    import struct
    from kaldi_amd import io as kio, synth
    from kaldi_amd._lib import KamdError
    g = synth.make_random_graph(num_states=20, num_labels=6, seed=4)
    p = tmp_path / "g.fst"
    kio.write_openfst(p, g, "const")
    raw = bytearray(p.read_bytes())
    # header: int32 magic | str fsttype | str arctype | int32 version | int32 flags | ...
    pos = 4
    for _ in range(2):
        n = struct.unpack_from("<i", raw, pos)[0]
        pos += 4 + n
    ver_pos, flag_pos = pos, pos + 4
    for patch, what in (((flag_pos, 8), "flags"), ((ver_pos, 7), "version"), ((0, 12345), "magic")):
        bad = bytearray(raw)
        struct.pack_into("<i", bad, patch[0], patch[1])
        q = tmp_path / ("bad_%s.fst" % what)
        q.write_bytes(bytes(bad))
        with pytest.raises(KamdError, match=what):
            kio.read_openfst(q)
    assert kio.read_openfst(p).num_states == g.num_states
