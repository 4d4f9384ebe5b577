"""ConstArpaLm, the ARPA reader and lattice-lmrescore-const-arpa, PINNED by the reference's own known answers restated as
data: the n-grams (with their line numbers) of lm/arpa-file-parser-test.cc:160-260, the two sentence scores of
lm/arpa-lm-compiler-test.cc:225-226 on lm/test_data/input.arpa (copied as data to tests/golden/lm/), its
missing-<s> failure case (:228), and the on-disk layout rules of lm/const-arpa-lm.cc:330-560."""
import math
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from kaldi_amd import constarpa, latbin
from kaldi_amd._lib import KamdError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LM = os.path.join(ROOT, "tests", "golden", "lm")
LN10 = math.log(10.0)

INTEGER_LM = ("\\data\\\nngram 1=4\nngram 2=2\nngram 3=2\n\n\\1-grams:\n-5.2\t4\t-3.3\n-3.4\t5\n0\t1\t-2.5\n-4.3\t2\n\n"
              "\\2-grams:\n-1.4\t4 5\t-3.2\n-1.3\t1 4\t-4.2\n\n\\3-grams:\n-0.3\t1 4 5\n-0.2\t4 5 2\n\n\\end\\")
# arpa-file-parser-test.cc:176-187 (line, logprob, words, backoff), log10 values
EXPECT_INTEGER = [(7, -5.2, [4], -3.3), (8, -3.4, [5], 0.0), (9, 0.0, [1], -2.5), (10, -4.3, [2], 0.0),
                  (13, -1.4, [4, 5], -3.2), (14, -1.3, [1, 4], -4.2), (17, -0.3, [1, 4, 5], 0.0), (18, -0.2, [4, 5, 2], 0.0)]
SYMBOLIC_LM = ("We also allow random text coming before the \\data\\\nsection marker. Even this is ok:\n\n\\1-grams:\n\n"
               "and should be ignored before the \\data\\ marker\nis seen alone by itself on a line.\n\n\\data\\\nngram 1=4\nngram 2=2\n"
               "ngram 3=2\n\n\\1-grams: \n-5.2\ta\t-3.3\n-3.4\t\u03b2\n0.0\t<s>\t-2.5\n-4.3\t</s>\n\n\\2-grams:\t\n-1.5\ta \u03b2\t-3.2\n"
               "-1.3\t<s> a\t-4.2\n\n\\3-grams:\n-0.3\t<s> a \u03b2\n-0.2\t<s> a </s>\n\\end\\")
# :247-258 with the symbol table <eps> 0, <s> 1, </s> 2, <unk> 3, a 4 and beta added as 5
EXPECT_SYMBOLIC = [(15, -5.2, [4], -3.3), (16, -3.4, [5], 0.0), (17, 0.0, [1], -2.5), (18, -4.3, [2], 0.0),
                   (21, -1.5, [4, 5], -3.2), (22, -1.3, [1, 4], -4.2), (25, -0.3, [1, 4, 5], 0.0), (26, -0.2, [1, 4, 2], 0.0)]


def check_ngrams(got, want):
    assert len(got) == len(want)
    for (line, words, lp, bo), (eline, elp, ewords, ebo) in zip(got, want):
        assert line == eline and words == ewords
        assert lp == pytest.approx(elp * LN10, rel=1e-6, abs=1e-7) and bo == pytest.approx(ebo * LN10, rel=1e-6, abs=1e-7)


def test_arpa_reader_known_answers(tmp_path):
    p = tmp_path / "int.arpa"
    p.write_text(INTEGER_LM)
    counts, ngrams = constarpa.parse_arpa(p)
    assert counts == [4, 2, 2]
    check_ngrams(ngrams, EXPECT_INTEGER)
    q = tmp_path / "sym.arpa"
    q.write_bytes(SYMBOLIC_LM.encode("utf-8"))
    w = tmp_path / "words.txt"
    w.write_bytes("<eps> 0\n<s> 1\n</s> 2\n<unk> 3\na 4\n\u03b2 5\n".encode("utf-8"))
    counts, ngrams = constarpa.parse_arpa(q, w)
    assert counts == [4, 2, 2]
    check_ngrams(ngrams, EXPECT_SYMBOLIC)
    w2 = tmp_path / "words_no_beta.txt"                     # ReadSymbolicLmNoOovImpl: an unknown word is an error
    w2.write_text("<eps> 0\n<s> 1\n</s> 2\n<unk> 3\na 4\n")
    with pytest.raises(KamdError, match="not in symbol table"):
        constarpa.parse_arpa(q, w2)
    bad = tmp_path / "eps.arpa"
    bad.write_text(INTEGER_LM.replace("-3.4\t5", "-3.4\t0"))
    with pytest.raises(KamdError, match="epsilon"):
        constarpa.parse_arpa(bad)
    many = tmp_path / "many.arpa"
    many.write_text(INTEGER_LM.replace("ngram 2=2", "ngram 2=1"))
    with pytest.raises(KamdError, match="saw more"):
        constarpa.parse_arpa(many)


@pytest.fixture()
def words_txt(tmp_path):
    p = tmp_path / "words.txt"
    p.write_text("<eps> 0\n<s> 1\n</s> 2\na 3\nb 4\n")
    return p


def test_sentence_scores_of_the_reference(words_txt):
    """arpa-lm-compiler-test.cc:225-226: ScoringTest("test_data/input.arpa", "b b b a", 59.2649) and ("a b", 4.36082):
    the cost of the sentence through the LM FST, back-off included."""
    lm = constarpa.ConstArpaLm.build(os.path.join(LM, "input.arpa"), 1, 2, -1, words_txt)
    assert (lm.order, lm.num_words, lm.bos, lm.eos, lm.unk) == (3, 5, 1, 2, -1)
    assert lm.sentence_cost([4, 4, 4, 3]) == pytest.approx(59.2649, rel=1e-5)
    assert lm.sentence_cost([3, 4]) == pytest.approx(4.36082, rel=1e-5)
    # the pieces: explicit trigram, and the back-off chain of an unseen history
    assert lm.GetNgramLogprob(4, [1, 3]) == pytest.approx(-0.34958 * LN10, rel=1e-6)
    assert lm.GetNgramLogprob(4, [1]) == pytest.approx((-2.5 - 3.456783) * LN10, rel=1e-6)
    assert lm.GetNgramLogprob(2, [7, 7, 3]) == pytest.approx((-3.3 - 4.333333) * LN10, rel=1e-6)   # history truncated to order - 1, unknown words dropped
    # no <unk>: a word outside the LM gets FLT_MIN from the unigram case (const-arpa-lm.cc:790-796) ...
    assert lm.GetNgramLogprob(9, []) == constarpa.FLT_MIN
    # ... which the recursion ADDS to the back-off weights of a non-empty history (:817): the sum is the back-off weight
    # alone, not FLT_MIN, so ConstArpaLmDeterministicFst::GetArc's test (:1027) lets the word through.  Restated as is.
    assert lm.GetNgramLogprob(9, [1]) == pytest.approx(-2.5 * LN10, rel=1e-6)


def test_backoff_coverage_files_build_and_score(words_txt, tmp_path):
    """the reference's CoverageTest files (arpa-lm-compiler-test.cc:221-223): every random sentence gets a finite score.
    (missing_bos.arpa, :228, is an ArpaLmCompiler = G.fst failure: ConstArpaLm only requires 0 < <s> < num_words, :660-667;
    with a <s> id beyond the vocabulary the build is refused.)"""
    w = tmp_path / "w3.txt"
    w.write_text("<eps> 0\n<s> 1\n</s> 2\na 3\nb 4\nc 5\n")
    with pytest.raises(KamdError, match="<s>"):
        constarpa.ConstArpaLm.build(os.path.join(LM, "missing_bos.arpa"), 9, 2, -1, w)
    with pytest.raises(KamdError, match="BOS and EOS"):
        constarpa.ConstArpaLm.build(os.path.join(LM, "input.arpa"), 1, 1, -1, words_txt)
    # an n-gram whose history n-gram is absent: ConstArpaLmBuilder::ConsumeNGram's KALDI_ERR (:309-315), unlike ArpaLmCompiler
    with pytest.raises(KamdError, match="does not have a parent model"):
        constarpa.ConstArpaLm.build(os.path.join(LM, "missing_backoffs.arpa"), 1, 2, -1, w)
    with pytest.raises(KamdError, match="does not have a parent model"):
        constarpa.ConstArpaLm.build(os.path.join(LM, "unused_backoffs.arpa"), 1, 2, -1, w)
    lm = constarpa.ConstArpaLm.build(os.path.join(LM, "input.arpa"), 1, 2, -1, words_txt)
    rng = np.random.default_rng(0)
    for _ in range(50):                                      # CoverageTest: random sentences all get a score
        sent = rng.integers(3, 5, rng.integers(1, 9)).tolist()
        assert math.isfinite(lm.sentence_cost(sent))


def test_carpa_file_layout_and_round_trip(words_txt, tmp_path):
    lm = constarpa.ConstArpaLm.build(os.path.join(LM, "input.arpa"), 1, 2, -1, words_txt)
    p = tmp_path / "G.carpa"
    lm.write(p)
    raw = p.read_bytes()
    assert raw[:2] == b"\0B" and raw[2:17] == b"<ConstArpaLm> <" and raw.endswith(b"</LmOverflow> </ConstArpaLm> ")
    # <LmInfo>: four int32 with their size byte
    i = raw.index(b"<LmInfo> ") + 9
    vals = [struct.unpack_from("<bi", raw, i + 5 * k) for k in range(4)]
    assert [v[0] for v in vals] == [4, 4, 4, 4] and [v[1] for v in vals] == [1, 2, -1, 3]
    # <LmStates>: int64 count; states that exist: the 4 unigrams (<s>: 1 child, a: 1 child, b, </s>: leaves but unigrams)
    # and "<s> a", "a b" (order 2, children of the final order) = 3 + 2 + 3 + 2 + 3 + 3 + (3 + 2) + (3 + 2) ints
    j = raw.index(b"<LmStates> ") + 11
    assert raw[j] == 8
    n = struct.unpack_from("<q", raw, j + 1)[0]
    assert n == lm.lm_states_size == 3 + 2 + 3 + 2 + 3 + 3 + 5 + 5
    lm2 = constarpa.ConstArpaLm.read(p)
    for sent in ([4, 4, 4, 3], [3, 4], [3], [4, 3, 4, 3]):
        assert lm2.sentence_cost(sent) == lm.sentence_cost(sent)
    q = tmp_path / "G2.carpa"
    lm2.write(q)
    assert q.read_bytes() == raw
    # the old on-disk format (every value with its size byte, const-arpa-lm.cc:670-715)
    states = np.frombuffer(raw, "<i4", n, j + 9)
    k = raw.index(b"<LmUnigram> ") + 12
    nw = struct.unpack_from("<bi", raw, k)[1]
    uni = np.frombuffer(raw, "<i8", nw, k + 5)
    old = b"\0B" + b"".join(struct.pack("<bi", 4, v) for v in (1, 2, -1, 3)) + struct.pack("<bi", 4, n)
    old += b"".join(struct.pack("<bi", 4, int(v)) for v in states) + struct.pack("<bi", 4, nw)
    old += b"".join(struct.pack("<bq", 8, int(v)) for v in uni) + struct.pack("<bi", 4, 0)
    o = tmp_path / "old.carpa"
    o.write_bytes(old)
    lm3 = constarpa.ConstArpaLm.read(o)
    assert lm3.sentence_cost([4, 4, 4, 3]) == lm.sentence_cost([4, 4, 4, 3])
    t = tmp_path / "trunc.carpa"
    t.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(KamdError):
        constarpa.ConstArpaLm.read(t)
    # corrupt <LmStates> payloads are errors of the read, not out-of-bounds accesses at rescoring time: a child count that
    # runs past the block, a child offset that leaves it, an overflow index with no entry, a section larger than the file
    st = states.copy()
    roots = [int(a) - 1 for a in uni if a > 0]
    with_children = [r for r in roots if st[r + 2] > 0]
    assert with_children

    def rewrite(mutate, header_n=None):
        s2 = st.copy()
        mutate(s2)
        blob = raw[:j + 1] + struct.pack("<q", n if header_n is None else header_n) + s2.tobytes() + raw[j + 9 + 4 * n:]
        f = tmp_path / "bad.carpa"
        f.write_bytes(blob)
        return f

    r0 = with_children[0]
    for mutate, msg in ((lambda a: a.__setitem__(r0 + 2, 1 << 20), "children run past"),
                        (lambda a: a.__setitem__(r0 + 4, ((1 << 24) * 2) | 1), "outside <LmStates>"),
                        (lambda a: a.__setitem__(r0 + 4, -((5 * 2) | 1)), "overflow index")):
        with pytest.raises(KamdError, match=msg):
            constarpa.ConstArpaLm.read(rewrite(mutate))
    with pytest.raises(KamdError, match="corrupt <LmStates> header"):
        constarpa.ConstArpaLm.read(rewrite(lambda a: None, header_n=1 << 39))


def _sausage(paths):
    """a compact lattice that is a union of word sequences with given (graph, acoustic) costs per word"""
    lat = latbin.Lat(0)
    lat.add_state()
    for words, g, a in paths:
        cur = 0
        for k, w in enumerate(words):
            nxt = lat.add_state()
            lat.arcs[cur].append((nxt, w, np.float32(g), np.float32(a), [10 * w + k, 10 * w + k]))
            cur = nxt
        lat.final[cur] = (np.float32(0.5), np.float32(0.0), [])
    return lat


def test_lattice_rescoring_adds_the_lm_cost_to_every_path(words_txt):
    lm = constarpa.ConstArpaLm.build(os.path.join(LM, "input.arpa"), 1, 2, -1, words_txt)
    paths = [([3, 4], 1.0, 2.0), ([4, 4, 4, 3], 0.25, 1.0), ([3, 3], 2.0, 0.5)]
    lat = _sausage(paths)
    out = lm.rescore(lat, 1.0)
    assert out is not None

    def all_paths(L):
        res = {}
        def walk(s, words, g, a, tids):
            if L.final[s] is not None:
                fg, fa, ft = L.final[s]
                res[tuple(words)] = (g + float(fg), a + float(fa), tuple(tids + list(ft)))
            for d, wd, ag, aa, t in L.arcs[s]:
                walk(d, words + ([wd] if wd else []), g + float(ag), a + float(aa), tids + list(t))
        walk(L.start, [], 0.0, 0.0, [])
        return res

    before, after = all_paths(lat), all_paths(out)
    assert set(before) == set(after)
    for w in before:
        assert after[w][0] == pytest.approx(before[w][0] + lm.sentence_cost(list(w)), rel=1e-5)      # graph cost + LM cost
        assert after[w][1] == pytest.approx(before[w][1], rel=1e-6) and after[w][2] == before[w][2]   # acoustics and alignment untouched
    # --lm-scale=-1 then +1 with the same LM is the identity on path costs (how recipes swap G.fst for G.carpa)
    back = lm.rescore(out, -1.0)
    again = all_paths(back)
    for w in before:
        assert again[w][0] == pytest.approx(before[w][0], abs=2e-4)
    # a word outside the LM after <s>: scored with the back-off weight of <s> alone (the reference's behaviour, see above)
    oov = all_paths(lm.rescore(_sausage([([9], 1.0, 1.0), ([3], 1.0, 1.0)]), 1.0))
    assert set(oov) == {(9,), (3,)}
    assert oov[(9,)][0] == pytest.approx(1.0 + 0.5 + lm.sentence_cost([9]), rel=1e-5)


def test_command_line_tools(words_txt, tmp_path):
    """arpa-to-const-arpa | lattice-lmrescore-const-arpa | lattice-best-path on archives"""
    carpa = tmp_path / "G.carpa"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "arpa_to_const_arpa.py"), "--bos-symbol=1", "--eos-symbol=2",
                        "--read-symbol-table=" + str(words_txt), os.path.join(LM, "input.arpa"), str(carpa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "arpa_to_const_arpa.py"), os.path.join(LM, "input.arpa"), str(carpa)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "BOS and EOS" in r.stderr
    lat = _sausage([([3, 4], 1.0, 2.0), ([4, 4, 4, 3], 0.25, 1.0)])
    ark = tmp_path / "lat.ark"
    ark.write_bytes(b"utt1 " + latbin.compact_bytes(lat, True))
    out = tmp_path / "out.ark"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lattice_lmrescore_const_arpa.py"), "--lm-scale=1.0", "ark:" + str(ark),
                        str(carpa), "ark:" + str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Done 1 lattices, failed for 0" in r.stderr
    (key, got), = list(latbin.read_lattices("ark:" + str(out)))
    assert key == "utt1"
    lm = constarpa.ConstArpaLm.read(carpa)
    words, ali, g, a = latbin.best_path(got)
    # "a b": 2*1.0 + 0.5 + 4.36 graph, 4.0 acoustic = 10.86; "b b b a": 4*0.25 + 0.5 + 59.26, 4.0 -> the short sentence wins
    assert words == [3, 4] and g == pytest.approx(2.5 + lm.sentence_cost([3, 4]), rel=1e-5) and a == pytest.approx(4.0)
