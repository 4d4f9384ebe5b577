"""Table / command-line layer.  The classification and option-parsing cases are the reference's
own known-answer tests (util/kaldi-io-test.cc:31-69, util/kaldi-table-test.cc:91-326,
util/parse-options-test.cc:45-296), restated as data; the readers and writers are checked by
round trips through files, script files with offsets, pipes and --config files."""
import gzip
import os
import wave

import numpy as np
import pytest

from kaldi_amd import table as T
from kaldi_amd._lib import KamdError

RX = [("", T.RX_STDIN), (" ", T.RX_NONE), (" a ", T.RX_NONE), ("a ", T.RX_NONE), ("a", T.RX_FILE), ("-", T.RX_STDIN),
      ("b|", T.RX_PIPE), ("|b", T.RX_NONE), ("b c|", T.RX_PIPE), (" b c|", T.RX_PIPE), ("a b c:123", T.RX_OFFSET_FILE),
      ("a b c:3", T.RX_OFFSET_FILE), ("a b c:", T.RX_FILE), ("a b c/3", T.RX_FILE), ("ark,s,cs:a b c", T.RX_NONE),
      ("scp:a b c", T.RX_NONE)]
WX = [("", T.WX_STDOUT), (" ", T.WX_NONE), (" a ", T.WX_NONE), ("a ", T.WX_NONE), ("a", T.WX_FILE), ("-", T.WX_STDOUT),
      ("b|", T.WX_NONE), ("|b", T.WX_PIPE), ("| b ", T.WX_PIPE), ("b c|", T.WX_NONE), ("a b c:123", T.WX_NONE),
      ("ark,s,cs:a b c", T.WX_NONE), ("scp:a b c", T.WX_NONE), ("a b c:3", T.WX_NONE), ("a b c:", T.WX_FILE),
      ("a b c/3", T.WX_FILE)]


@pytest.mark.parametrize("name,want", RX)
def test_classify_rxfilename(name, want):
    assert T.classify_rxfilename(name) == want


@pytest.mark.parametrize("name,want", WX)
def test_classify_wxfilename(name, want):
    assert T.classify_wxfilename(name) == want


def test_classify_wspecifier_known_answers():
    c = T.classify_wspecifier
    assert c("b,ark:foo|")[:3] == (T.ARCHIVE, "foo|", "") and c("b,ark:foo|")[3]["binary"]
    assert c("t,ark:foo|")[:3] == (T.ARCHIVE, "foo|", "") and not c("t,ark:foo|")[3]["binary"]
    assert c("t,scp:a b c d")[:3] == (T.SCRIPT, "", "a b c d") and not c("t,scp:a b c d")[3]["binary"]
    assert c("t,ark,scp:a b,c,d")[:3] == (T.BOTH, "a b", "c,d")
    assert c("")[0] == T.NO_SPECIFIER
    assert c(" t,ark:boo")[0] == T.NO_SPECIFIER          # leading space
    assert c("t,ark:boo ")[0] == T.NO_SPECIFIER          # trailing space
    assert c("b,ark,scp:,")[:3] == (T.BOTH, "", "") and c("b,ark,scp:,")[3]["binary"]
    assert c("f,b,ark,scp:,")[3] == {"binary": True, "flush": True, "permissive": False}
    assert c("nf,b,ark,scp:,")[3] == {"binary": True, "flush": False, "permissive": False}
    assert c("scp,ark:a,b")[0] == T.NO_SPECIFIER         # only "ark,scp" is allowed


def test_classify_rspecifier_known_answers():
    c = T.classify_rspecifier
    for s in ("ark:foo|", "b,ark:foo|", "ark,b:foo|"):
        assert c(s)[:2] == (T.ARCHIVE, "foo|")
    assert c("scp,b:foo|")[:2] == (T.SCRIPT, "foo|")
    assert c("scp,scp,b:foo|")[:2] == (T.NO_SPECIFIER, "")
    assert c("ark,scp,b:foo|")[:2] == (T.NO_SPECIFIER, "")
    assert c("scp,o:foo|")[2]["once"] and not c("scp,no:foo|")[2]["once"]
    t, f, o = c("s,scp,no:foo|")
    assert (t, f) == (T.SCRIPT, "foo|") and o["sorted"] and not o["once"]
    assert c("scp:")[:2] == (T.SCRIPT, "")
    for s in ("", "scp", "ark", "ark:foo "):
        assert c(s)[0] == T.NO_SPECIFIER
    assert c("b,scp:a")[:2] == (T.SCRIPT, "a") and c("t,scp:a")[:2] == (T.SCRIPT, "a")
    assert c("b,ark:a")[:2] == (T.ARCHIVE, "a") and c("t,ark:a")[:2] == (T.ARCHIVE, "a")
    assert c("ark,s,cs:x")[2] == {"once": False, "sorted": True, "called_sorted": True, "permissive": False, "background": False}


def _po():
    po = T.ParseOptions("my usage msg")
    return po


def test_parse_options_known_answers(tmp_path):
    po = _po()
    po.register("i", str, "default_for_str"); po.register("num", int, 1); po.register("unum", "uint", 2)
    assert po.read(["program_name", "--unum=5", "--num=3", "--i=boo", "a", "b", "c"]) == ["a", "b", "c"]
    assert po.num_args() == 3 and po.get_arg(1) == "a" and po.get_arg(3) == "c"
    assert (po["unum"], po["num"], po["i"]) == (5, 3, "boo")
    po2 = _po()
    po2.register("To_Be_Normalized", str, "d"); po2.register("i", str, "x")
    assert po2.read(["p", "--i=foo", "--to-be-NORMALIZED=test", "c"]) == ["c"]
    assert po2["to-be-normalized"] == "test" and po2["i"] == "foo"
    # prefixed registration, recursively (ParseOptions(prefix, other))
    po3 = _po()
    ro3 = T.ParseOptions("prefix", po3)
    so3 = T.ParseOptions("prefix2", ro3)
    po3.register("str", str, ""); po3.register("num", int, 0)
    ro3.register("unum", "uint", 0); ro3.register("str", str, "")
    so3.register("unum", "uint", 0)
    ro3.register("my-bool", bool, True); ro3.register("my-str", str, "default dummy string")
    assert po3.read(["program_name", "--prefix.unum=5", "--num=3", "--prefix.str=foo", "--str=bar", "--prefix.my-bool=false",
                     "--prefix.my-str=baz", "--prefix.prefix2.unum=42", "a", "b"]) == ["a", "b"]
    assert ro3["unum"] == 5 and so3["unum"] == 42 and po3["num"] == 3 and ro3["str"] == "foo" and po3["str"] == "bar"
    assert ro3["my-bool"] is False and ro3["my-str"] == "baz"

    def one(typ, default, arg):
        p = _po()
        p.register("option", typ, default)
        p.read(["program_name", arg])
        return p["option"]
    with pytest.raises(KamdError):
        one(bool, False, "--option=")
    assert one(bool, False, "--option") is True
    with pytest.raises(KamdError):
        one(str, "", "--option")
    assert one(str, "foo", "--option=") == ""
    assert one(int, 32, "--option=8") == 8
    assert one(float, 32.0, "--option=8.5") == 8.5
    assert one(str, "foo", "--option=bar") == "bar"
    for typ, arg in ((float, "--option=foo"), (int, "--option=foo"), (int, "--option=12xyz"), ("uint", "--option=-13"),
                     (bool, "--option=foo"), (int, "--=8")):
        with pytest.raises(KamdError):
            one(typ, 0, arg)
    for argv, want_args, want in ((["p", "--unum=6", "--", "a", "b"], ["a", "b"], 6), (["p", "--unum=7", "--"], [], 7),
                                  (["p", "--unum=8", "--", "--foo=8"], ["--foo=8"], 8)):
        p = _po()
        p.register("unum", "uint", 2)
        assert p.read(argv) == want_args and p["unum"] == want
    # --config (parse-options.cc:459-495): comments, blank lines, '--x=y' lines; the command line wins
    cfg = tmp_path / "decode.conf"
    cfg.write_text("# decoder\n--beam=13.0   # tighter\n\n--max_active=5000\n--Allow-Partial\n")
    p = _po()
    p.register("beam", float, 16.0); p.register("max-active", int, 2**31 - 1); p.register("allow-partial", bool, False)
    assert p.read(["p", "--config=%s" % cfg, "--beam=11", "x"]) == ["x"]
    assert (p["beam"], p["max-active"], p["allow-partial"]) == (11.0, 5000, True)
    cfg.write_text("beam=13\n")
    with pytest.raises(KamdError):
        _po().read(["p", "--config=%s" % cfg])
    with pytest.raises(KamdError):
        p.read(["p", "--no-such-option=1"])


def test_matrix_tables_through_ark_scp_offsets_and_pipes(tmp_path):
    rng = np.random.default_rng(0)
    mats = {"utt%02d" % i: rng.standard_normal((rng.integers(1, 9), 5)).astype(np.float32) for i in range(6)}
    ark, scp = str(tmp_path / "f.ark"), str(tmp_path / "f.scp")
    with T.TableWriter("ark,scp:%s,%s" % (ark, scp), "matrix") as w:
        for k in sorted(mats):
            w.write(k, mats[k])
    lines = open(scp).read().splitlines()
    assert len(lines) == 6 and lines[0].split()[1].startswith(ark + ":")
    raw = open(ark, "rb").read()
    off = int(lines[2].rsplit(":", 1)[1])
    assert raw[off:off + 2] == b"\0B" and raw[off - 6:off] == b"utt02 "       # the offset points at the object
    for spec in ("ark:" + ark, "scp:" + scp, "ark,s,cs:cat %s |" % ark, "scp,p:" + scp):
        got = dict(T.SequentialTableReader(spec, "matrix"))
        assert sorted(got) == sorted(mats)
        for k in mats:
            np.testing.assert_array_equal(got[k], mats[k])
    ra = T.RandomAccessTableReader("scp:" + scp, "matrix")
    assert "utt03" in ra and "nope" not in ra
    np.testing.assert_array_equal(ra["utt03"], mats["utt03"])
    ra2 = T.RandomAccessTableReader("ark:" + ark, "matrix")
    np.testing.assert_array_equal(ra2.value("utt05"), mats["utt05"])
    # gzip through pipes both ways
    gz = str(tmp_path / "f.ark.gz")
    with T.TableWriter("ark,t:| gzip -c > %s" % gz, "matrix") as w:
        for k in sorted(mats):
            w.write(k, mats[k])
    assert gzip.open(gz).read().startswith(b"utt00  [\n")
    got = dict(T.SequentialTableReader("ark:gunzip -c %s |" % gz, "matrix"))
    for k in mats:
        np.testing.assert_allclose(got[k], mats[k], rtol=1e-6)
    # sorted option is enforced; a failing command is an error; permissive scp skips bad entries
    with T.TableWriter("ark:" + ark, "matrix") as w:
        w.write("b", mats["utt00"]); w.write("a", mats["utt01"])
    with pytest.raises(KamdError):
        list(T.SequentialTableReader("ark,s:" + ark, "matrix"))
    assert [k for k, _ in T.SequentialTableReader("ark:" + ark, "matrix")] == ["b", "a"]
    with pytest.raises(KamdError):
        list(T.SequentialTableReader("ark:false |", "matrix"))
    open(scp, "w").write("x %s:9\ny /no/such/file\n" % ark)
    with pytest.raises(KamdError):
        list(T.SequentialTableReader("scp:" + scp, "matrix"))
    assert list(T.SequentialTableReader("scp,p:" + scp, "matrix")) == []
    for bad in ("k\n", "\n", "k  \n"):
        open(scp, "w").write(bad)
        with pytest.raises(KamdError):
            T.read_script_file(scp)
    with pytest.raises(KamdError):
        T.TableWriter("ark:x|", "matrix")
    with pytest.raises(KamdError):
        T.SequentialTableReader("feats.ark", "matrix")


@pytest.mark.parametrize("binary", [True, False])
def test_int32_and_wave_tables(tmp_path, binary, monkeypatch):
    spool = tmp_path / "spool"                       # the spooled pipes of THIS test only (other workers share /tmp)
    spool.mkdir()
    monkeypatch.setenv("TMPDIR", str(spool))
    vecs = {"a": np.array([1, 2, 3], np.int32), "b": np.zeros(0, np.int32), "c": np.array([-7, 2**31 - 1], np.int32)}
    ark, scp = str(tmp_path / "v.ark"), str(tmp_path / "v.scp")
    with T.TableWriter("ark%s,scp:%s,%s" % ("" if binary else ",t", ark, scp), "int32") as w:
        for k, v in vecs.items():
            w.write(k, v)
        with pytest.raises(KamdError):
            w.write("bad key", vecs["a"])
    for spec in ("ark:" + ark, "scp:" + scp):
        got = dict(T.SequentialTableReader(spec, "int32"))
        assert list(got) == list(vecs)
        for k in vecs:
            np.testing.assert_array_equal(got[k], vecs[k])
    # wav.scp with a plain file and a command
    pcm = (np.arange(800) * 37 % 2000 - 1000).astype("<i2")
    wp = str(tmp_path / "a.wav")
    with wave.open(wp, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    open(scp, "w").write("u1 %s\nu2 cat %s |\n" % (wp, wp))
    got = list(T.SequentialTableReader("scp:" + scp, "wave"))
    assert [k for k, _ in got] == ["u1", "u2"]
    for _, (sf, data) in got:
        assert sf == 16000.0
        np.testing.assert_array_equal(data[0], pcm.astype(np.float32))
    assert not [p for p in os.listdir(os.environ.get("TMPDIR", "/tmp")) if p.startswith("kamd_rx_")]


def test_vector_and_token_tables(tmp_path):
    """per-utterance / per-speaker i-vector archives (binary FV, DV, text) and utt2spk-style token lines"""
    import struct
    ark = str(tmp_path / "iv.ark")
    v1, v2, v3 = np.array([1.5, -2, 3], np.float32), np.array([0.25, 4], np.float64), np.array([7, 8.5, -9, 10], np.float32)
    with open(ark, "wb") as f:
        f.write(b"spk1 \0BFV \x04" + struct.pack("<i", 3) + v1.tobytes())
        f.write(b"spk2 \0BDV \x04" + struct.pack("<i", 2) + v2.tobytes())
        f.write(b"spk3  [ 7 8.5 -9 10 ]\n")
    got = list(T.SequentialTableReader("ark:" + ark, "vector"))
    assert [k for k, _ in got] == ["spk1", "spk2", "spk3"]
    for (_, g), w in zip(got, (v1, v2, v3)):
        np.testing.assert_array_equal(g, w.astype(np.float32))
    off2 = open(ark, "rb").read().index(b"spk2 ") + 5
    open(tmp_path / "iv.scp", "w").write("a %s:%d\n" % (ark, off2))
    np.testing.assert_array_equal(T.RandomAccessTableReader("scp:%s" % (tmp_path / "iv.scp"), "vector")["a"], v2.astype(np.float32))
    (tmp_path / "utt2spk").write_text("utt1 spk1\nutt2 spk1\nutt3 spk2\n")
    assert dict(T.SequentialTableReader("ark:%s" % (tmp_path / "utt2spk"), "tokens")) == {"utt1": ["spk1"], "utt2": ["spk1"], "utt3": ["spk2"]}


def test_double_matrix_tables_round_trip(tmp_path):
    """CMVN statistics are Matrix<double> tables (DoubleMatrixWriter / RandomAccessDoubleMatrixReader): binary DM, text,
    ark+scp, and float (FM) archives read into doubles."""
    from kaldi_amd import io as kio
    rng = np.random.default_rng(0)
    mats = {"spk%d" % i: rng.standard_normal((2, 14)) * 1e6 + np.pi for i in range(3)}
    for spec in ("ark:%s" % (tmp_path / "a.ark"), "ark,t:%s" % (tmp_path / "t.ark"),
                 "ark,scp:%s,%s" % (tmp_path / "b.ark", tmp_path / "b.scp")):
        with T.TableWriter(spec, "dmatrix") as w:
            for k, m in mats.items():
                w.write(k, m)
    assert open(tmp_path / "a.ark", "rb").read().startswith(b"spk0 \0BDM \x04\x02\x00\x00\x00\x04\x0e\x00\x00\x00")
    assert open(tmp_path / "t.ark").read().startswith("spk0  [\n  ")
    for spec in ("ark:%s" % (tmp_path / "a.ark"), "ark:%s" % (tmp_path / "t.ark"), "scp:%s" % (tmp_path / "b.scp")):
        got = dict(T.SequentialTableReader(spec, "dmatrix"))
        assert list(got) == list(mats)
        for k in mats:
            assert got[k].dtype == np.float64
            np.testing.assert_array_equal(got[k], mats[k])                   # exact: doubles all the way (text: %.17g)
    ra = T.RandomAccessTableReader("scp:%s" % (tmp_path / "b.scp"), "dmatrix")
    np.testing.assert_array_equal(ra["spk2"], mats["spk2"])
    kio.write_matrix_ark(tmp_path / "f.ark", "x", mats["spk1"].astype(np.float32), append=False)
    (k, m), = list(T.SequentialTableReader("ark:%s" % (tmp_path / "f.ark"), "dmatrix"))
    assert k == "x" and m.dtype == np.float64
    np.testing.assert_array_equal(m, mats["spk1"].astype(np.float32).astype(np.float64))


def test_background_reader_option(tmp_path):
    """`bg` (SequentialTableReaderBackgroundImpl, util/kaldi-table-inl.h): same objects in the same order, read one ahead
    by a second thread; its errors surface at the object they belong to; a consumer that stops early does not hang."""
    import threading
    rng = np.random.default_rng(3)
    mats = {"utt%02d" % i: rng.standard_normal((rng.integers(1, 9), 4)).astype(np.float32) for i in range(9)}
    ark, scp = str(tmp_path / "b.ark"), str(tmp_path / "b.scp")
    with T.TableWriter("ark,scp:%s,%s" % (ark, scp), "matrix") as w:
        for k in sorted(mats):
            w.write(k, mats[k])
    assert T.classify_rspecifier("ark,bg:" + ark)[2]["background"]
    n0 = threading.active_count()
    for spec in ("ark,bg:" + ark, "scp,bg:" + scp, "ark,s,cs,bg:cat %s |" % ark):
        got = list(T.SequentialTableReader(spec, "matrix"))
        assert [k for k, _ in got] == sorted(mats)
        for k, v in got:
            np.testing.assert_array_equal(v, mats[k])
    it = iter(T.SequentialTableReader("ark,bg:" + ark, "matrix"))
    assert next(it)[0] == "utt00"
    it.close()                                                   # early exit: the reading thread is released and joined
    assert threading.active_count() == n0
    lines = open(scp).read().splitlines()
    open(scp, "w").write("\n".join(lines[:3] + ["bad /no/such/file"] + lines[3:]) + "\n")
    seen = []
    with pytest.raises((KamdError, OSError)):
        for k, _ in T.SequentialTableReader("scp,bg:" + scp, "matrix"):
            seen.append(k)
    assert seen == sorted(mats)[:3]                              # everything before the bad entry was delivered
    assert [k for k, _ in T.SequentialTableReader("scp,bg,p:" + scp, "matrix")] == sorted(mats)
