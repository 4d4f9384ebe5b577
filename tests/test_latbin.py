"""lattice-scale | lattice-add-penalty | lattice-best-path | compute-wer on the host (kaldi_amd/latbin.py): the
scoring chain of local/score.sh over lattices written by this repo's own writers (C-ABI), both archive kinds
and both forms."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kaldi_amd import abi, latbin, synth, table
from kaldi_amd import io as kio
from kaldi_amd._lib import KamdError
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_edit_distance_counts_and_wer_lines():
    r = "the cat sat on the mat".split()
    assert latbin.edit_distance(r, r) == (0, 0, 0, 0)
    assert latbin.edit_distance(r, "the cat sat on mat".split()) == (1, 0, 1, 0)
    assert latbin.edit_distance(r, "the big cat sat on the mat".split()) == (1, 1, 0, 0)
    assert latbin.edit_distance(r, "the cat sit on the mat".split()) == (1, 0, 0, 1)
    assert latbin.edit_distance(r, [])[:1] + latbin.edit_distance(r, [])[2:3] == (6, 6)
    assert latbin.edit_distance([], r) == (6, 6, 0, 0)
    tot, i, d, s = latbin.edit_distance("a b c d".split(), "b x d e f".split())
    assert tot == i + d + s == 4
    ref = {"u1": r, "u2": "hello world".split(), "u3": ["x"]}
    hyp = {"u1": "the cat sit on the mat".split(), "u2": "hello world".split()}
    with pytest.raises(KamdError):
        latbin.compute_wer(ref, hyp)                                   # strict: u3 missing
    lines = latbin.compute_wer(ref, hyp, "present")
    assert lines == ("%WER 12.50 [ 1 / 8, 0 ins, 0 del, 1 sub ] [PARTIAL]", "%SER 50.00 [ 1 / 2 ]", "Scored 2 sentences, 1 not present in hyp.")
    lines = latbin.compute_wer(ref, hyp, "all")
    assert lines[0] == "%WER 22.22 [ 2 / 9, 0 ins, 1 del, 1 sub ] [PARTIAL]" and lines[1] == "%SER 66.67 [ 2 / 3 ]"


def _lattices(tmp_path):
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=1)
    cfg = abi.decoder_config_recipe(); cfg.lattice_beam = 6.0
    tp = np.zeros(g.tid2pdf.size, np.int32); tp[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)
    out = {}
    for i in range(3):
        ll, words, _ = synth.sample_utterance(g, n_words=3 + i, seed=10 + i, peak=3.0)
        d = orc.Decoder(g, cfg, 1); d.Decode(ll)
        lat = d.GetRawLattice()
        out["utt%d" % i] = (lat, kio.determinize_lattice(lat, cfg.lattice_beam, tp), lat.best_path())
    return out


@pytest.mark.parametrize("binary", [True, False])
def test_read_both_lattice_kinds_and_best_path(tmp_path, binary):
    lats = _lattices(tmp_path)
    raw, comp = str(tmp_path / "raw.ark"), str(tmp_path / "clat.ark")
    for i, (k, (lat, clat, _)) in enumerate(lats.items()):
        kio.write_lattice(raw, k, lat, binary=binary, append=i > 0)
        clat.write(comp, k, binary=binary, append=i > 0)
    for path in (raw, comp):
        got = dict(latbin.read_lattices("ark:" + path))
        assert list(got) == list(lats)
        for k, (lat, clat, bp) in lats.items():
            words, ali, gc, ac = latbin.best_path(got[k])
            assert words == bp["words"].tolist()
            assert abs((gc + ac) - (bp["graph_cost"] + bp["acoustic_cost"])) < 1e-3 * max(1.0, abs(gc + ac))
            assert ali == bp["alignment"].tolist()
    # re-serialise the compact lattices with the Python writer in the other form and read them back: same structure
    other = str(tmp_path / "again.ark")
    with table.TableWriter(("ark:" if not binary else "ark,t:") + other, "raw") as w:
        for k, lat in latbin.read_lattices("ark:" + comp):
            w.write(k, latbin.compact_bytes(lat, binary=not binary))
    a, b = dict(latbin.read_lattices("ark:" + comp)), dict(latbin.read_lattices("ark:" + other))
    for k in a:
        assert a[k].start == b[k].start and len(a[k].final) == len(b[k].final)
        for st in range(len(a[k].final)):
            fa, fb = a[k].final[st], b[k].final[st]
            assert (fa is None) == (fb is None)
            if fa is not None:
                assert fa[2] == fb[2] and np.allclose(fa[:2], fb[:2], rtol=1e-6, atol=1e-6)
            assert [(d, w, t) for d, w, _, _, t in a[k].arcs[st]] == [(d, w, t) for d, w, _, _, t in b[k].arcs[st]]


def test_score_chain_through_pipes(tmp_path):
    """lattice-scale --inv-acoustic-scale=12 | lattice-add-penalty --word-ins-penalty=0.5 | lattice-best-path | compute-wer"""
    lats = _lattices(tmp_path)
    comp = str(tmp_path / "clat.ark")
    for i, (k, (_, clat, _)) in enumerate(lats.items()):
        clat.write(comp, k, binary=True, append=i > 0)
    py = sys.executable
    rspec = "ark:%s %s/tools/lattice_scale.py --inv-acoustic-scale=12 ark:%s ark:- | %s %s/tools/lattice_add_penalty.py --word-ins-penalty=0.5 ark:- ark:- |" % (
        py, ROOT, comp, py, ROOT)
    r = subprocess.run([py, ROOT + "/tools/lattice_best_path.py", rspec, "ark,t:%s" % (tmp_path / "hyp.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    hyp = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in open(tmp_path / "hyp.txt")}
    for k, lat in latbin.read_lattices("ark:" + comp):
        want = latbin.best_path(latbin.add_penalty(latbin.scale(lat, acoustic_scale=1.0 / 12), 0.5))
        assert hyp[k] == want[0]
    # with the acoustics scaled down 12x and a penalty per word the path is not longer than the unscaled best path
    assert sum(len(v) for v in hyp.values()) <= sum(len(b["words"]) for _, _, b in lats.values())
    (tmp_path / "ref.txt").write_text("".join("%s %s\n" % (k, " ".join(str(w) for w in b["words"])) for k, (_, _, b) in lats.items()))
    r = subprocess.run([py, ROOT + "/tools/compute_wer.py", "--text", "--mode=present", "ark:%s" % (tmp_path / "ref.txt"),
                        "ark:%s" % (tmp_path / "hyp.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0].startswith("%WER ") and out[1].startswith("%SER ") and out[2] == "Scored 3 sentences, 0 not present in hyp."
    # scale matrix semantics
    lat = next(latbin.read_lattices("ark:" + comp))[1]
    g0, a0 = [(g, a) for arcs in lat.arcs for _, _, g, a, _ in arcs][0]
    lat2 = latbin.scale(next(latbin.read_lattices("ark:" + comp))[1], lm_scale=2.0, acoustic_scale=0.5, acoustic2lm_scale=0.25, lm2acoustic_scale=3.0)
    g1, a1 = [(g, a) for arcs in lat2.arcs for _, _, g, a, _ in arcs][0]
    assert abs(g1 - (2.0 * g0 + 0.25 * a0)) < 1e-5 and abs(a1 - (3.0 * g0 + 0.5 * a0)) < 1e-5


def test_installed_wrappers_run_from_path(tmp_path):
    """tools/install_wrappers.py: Kaldi's binary names on $PATH; every tool script it points at exists."""
    bindir = tmp_path / "bin"
    r = subprocess.run([sys.executable, ROOT + "/tools/install_wrappers.py", str(bindir)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    names = sorted(os.listdir(bindir))
    assert "nnet3-latgen-faster" in names and "compute-wer" in names and len(names) >= 18
    for n in names:
        body = open(bindir / n).read()
        script = [w for w in body.split() if w.endswith(".py")][0]
        assert os.path.exists(script), script
    (tmp_path / "ref").write_text("u1 a b c\n")
    (tmp_path / "hyp").write_text("u1 a x c\n")
    env = dict(os.environ, PATH=str(bindir) + os.pathsep + os.environ["PATH"])
    r = subprocess.run(["compute-wer", "--text", "--mode=present", "ark:%s" % (tmp_path / "ref"), "ark:%s" % (tmp_path / "hyp")],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("%WER 33.33 [ 1 / 3, 0 ins, 0 del, 1 sub ]"), (r.stdout, r.stderr)


def test_edit_distance_reference_known_answers():
    """util/edit-distance-test.cc:26-65 (totals as the pairs grow) and :112-144 (ins/del/sub split)"""
    from kaldi_amd.latbin import edit_distance
    a, b, totals = [], [], []
    steps = [("a", 1), ("b", 1), ("b", 2), ("a", 2), ("a", 3), ("a", 4), ("b", 4), ("a", 5), ("b", 6), ("a", 1), ("b", 1), ("b", 10)]
    check_after = {0: 1, 1: 0, 2: 1, 3: 0, 6: 1, 7: 2, 8: 2, 10: 2, 11: 3}
    assert edit_distance(a, b)[0] == 0
    for i, (which, v) in enumerate(steps):
        (a if which == "a" else b).append(v)
        if i in check_after:
            assert edit_distance(a, b)[0] == check_after[i], (i, a, b)
            assert edit_distance([str(x) for x in a], [str(x) for x in b])[0] == check_after[i]
    hyp, ref = [1, 3, 4, 5], [2, 3, 4, 5, 6, 7]
    assert edit_distance(ref, hyp) == (3, 0, 2, 1)
    assert edit_distance(hyp, ref) == (3, 2, 0, 1)
    assert edit_distance([1], [1]) == (0, 0, 0, 0)
    assert edit_distance([1, 3], [1, 2]) == (1, 0, 0, 1)


def _random_dag(rng, n_states, words):
    from kaldi_amd import latbin as lb
    L = lb.Lat(0)
    for _ in range(n_states):
        L.add_state()
    for s in range(n_states - 1):
        for _ in range(int(rng.integers(1, 4))):
            d = int(rng.integers(s + 1, n_states))
            L.arcs[s].append((d, int(rng.choice(words)), float(rng.uniform(0, 3)), float(rng.uniform(0, 3)), [s + 1]))
    for s in range(n_states):
        if s == n_states - 1 or rng.random() < 0.2:
            L.final[s] = (float(rng.uniform(0, 2)), 0.0, [])
    return L


def _all_paths(L):
    out = []

    def walk(s, words, cost):
        if L.final[s] is not None:
            out.append((list(words), cost + L.final[s][0] + L.final[s][1]))
        for d, w, g, a, t in L.arcs[s]:
            walk(d, words + [w] if w else words, cost + g + a)
    walk(L.start, [], 0.0)
    return out


def test_nbest_and_oracle_against_path_enumeration():
    """lattice-to-nbest's n cheapest paths and lattice-oracle's smallest edit distance, against a brute-force walk over every
    path of small random acyclic lattices (epsilon arcs included)."""
    import numpy as np
    from kaldi_amd import latbin as lb
    rng = np.random.default_rng(5)
    for trial in range(40):
        L = _random_dag(rng, int(rng.integers(3, 9)), [0, 1, 2, 3, 4])
        paths = sorted(_all_paths(L), key=lambda p: p[1])
        got = lb.nbest(L, 7)
        assert len(got) == min(7, len(paths))
        for (w, c), (w2, c2) in zip(got, paths):
            assert abs(c - c2) < 1e-9
        assert [round(c, 9) for _, c in got] == sorted(round(c, 9) for _, c in got)
        # every returned (words, cost) is a path of the lattice
        want = {(tuple(w), round(c, 6)) for w, c in paths}
        assert all((tuple(w), round(c, 6)) in want for w, c in got)
        ref = [int(x) for x in rng.integers(1, 5, int(rng.integers(0, 5)))]
        brute = min(lb.edit_distance(ref, w)[0] for w, _ in paths)
        assert lb.oracle_errors(L, ref) == brute


def test_bench_word_path_acceptance_on_a_determinized_lattice():
    """bench.py's `_clat_accepts` (the containment check of cpu_baseline.divergence_from_reference_search): a word sequence is
    a path of the determinized lattice iff start -> ... -> a final state spells it, epsilon-labelled arcs being free."""
    import sys
    sys.argv = ["bench.py"]
    import bench
    from kaldi_amd import abi

    class Clat:
        pass
    cl = Clat()
    cl.num_states, cl.start = 5, 0
    cl.final = np.full(10, np.inf, np.float32)
    cl.final[2 * 3] = 0.0
    cl.final[2 * 4] = 0.0
    arcs = np.zeros(6, abi.CLAT_ARC_DTYPE)
    for i, (s, d, lab) in enumerate([(0, 1, 7), (1, 2, 0), (2, 3, 9), (0, 2, 8), (1, 4, 5), (3, 4, 0)]):
        arcs[i]["src"], arcs[i]["dst"], arcs[i]["label"] = s, d, lab
    cl.arcs = arcs
    assert bench._clat_accepts(cl, [7, 9]) and bench._clat_accepts(cl, [8, 9]) and bench._clat_accepts(cl, [7, 5])
    assert not bench._clat_accepts(cl, [7]) and not bench._clat_accepts(cl, [9]) and not bench._clat_accepts(cl, [7, 9, 9])
    assert not bench._clat_accepts(cl, []) and not bench._clat_accepts(None, [7])
