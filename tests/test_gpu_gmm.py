"""BASELINE configs[0]: a monophone-GMM system over a tiny HCLG decoded with gmm-latgen-faster's flow (the egs/yesno
plumbing case: "the test set is perfectly recognized").  Synthetic stand-in for yesno: every pdf is a small diagonal GMM
with well separated means; features are sampled from the GMMs along a known word sequence."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kaldi_amd import abi, decoder, gmm, latbin, synth, table
from kaldi_amd import io as kio
from oracle import orc
from tests import mdl_writer
from tests.util import lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_system(U=12, dim=13, seed=0):
    rng = np.random.default_rng(seed)
    g = synth.make_hclg(num_units=U, vocab=30, n_hist=6, seed=4)
    W, MIV, IV, means = [], [], [], []
    for p in range(2 * U):
        M = int(rng.integers(1, 4))
        mu = rng.standard_normal((M, dim)) * 0.3 + rng.standard_normal(dim) * 3.0
        var = rng.uniform(0.5, 1.5, (M, dim))
        w = rng.dirichlet(np.full(M, 3.0))
        W.append(w.astype(np.float32)); MIV.append((mu / var).astype(np.float32)); IV.append((1.0 / var).astype(np.float32))
        means.append((mu, var, w))
    return g, gmm.AmDiagGmm(W, MIV, IV), means


def sample_feats(g, means, n_words, seed):
    _, words, pdfs = synth.sample_utterance(g, n_words=n_words, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    x = np.zeros((pdfs.size, means[0][0].shape[1]), np.float32)
    for t, p in enumerate(pdfs):
        mu, var, w = means[int(p)]
        m = rng.choice(len(w), p=w)
        x[t] = mu[m] + rng.standard_normal(mu.shape[1]) * np.sqrt(var[m])
    return x, words


def test_gmm_loglikes_and_perfect_recognition(tmp_path):
    g, am_native, means = make_system()
    # the model file's TransitionModel (tests/mdl_writer.py) numbers the two pdfs of a phone the other way round than the
    # synthetic graph does: give model pdf id2pdf[tid] the GMM of the graph's own pdf of that transition-id
    _, id2pdf_w, _ = mdl_writer.transition_model(12)
    perm = np.zeros(24, np.int64)
    for tid in range(1, id2pdf_w.size):
        perm[id2pdf_w[tid]] = g.tid2pdf[tid]
    am = gmm.AmDiagGmm([am_native.pdf(int(n))[0] for n in perm], [am_native.pdf(int(n))[1] for n in perm], [am_native.pdf(int(n))[2] for n in perm])
    dec_am = gmm.DecodableAmDiagGmmScaled(am)
    utts = {"utt%d" % i: sample_feats(g, means, 2 + i, 50 + i) for i in range(4)}
    x0 = utts["utt3"][0]
    for scale in (1.0, 0.1):
        got, want = dec_am.loglikes(x0, scale), orc.am_gmm_loglikes(am, x0, scale)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))
    # files: final.mdl (TransitionModel + AmDiagGmm), HCLG.fst, feats.ark
    tm, id2pdf, _ = mdl_writer.transition_model(12)
    with open(tmp_path / "final.mdl", "wb") as f:
        f.write(b"\0B" + tm)
        gmm.write_am_diag_gmm(f, am)
    am2, id2pdf2, tid_phone, tid2phone = gmm.read_gmm_mdl(tmp_path / "final.mdl")
    np.testing.assert_array_equal(id2pdf2, id2pdf)
    for k in ("mix_off", "weights", "means_invvars", "inv_vars", "gconsts"):
        np.testing.assert_array_equal(getattr(am2, k), getattr(am, k))
    kio.write_openfst(tmp_path / "HCLG.fst", g, "const")
    with table.TableWriter("ark:%s" % (tmp_path / "feats.ark"), "matrix") as w:
        for k, (x, _) in utts.items():
            w.write(k, x)
    r = subprocess.run([sys.executable, ROOT + "/tools/gmm_latgen_faster.py", "--beam=13", "--lattice-beam=6", "--acoustic-scale=0.1", "--max-active=7000",
                        str(tmp_path / "final.mdl"), str(tmp_path / "HCLG.fst"), "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "lat.ark"),
                        "ark,t:%s" % (tmp_path / "hyp.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Done 4 utterances, failed for 0" in r.stderr
    (tmp_path / "ref.txt").write_text("".join("%s %s\n" % (k, " ".join(str(w) for w in ws)) for k, (_, ws) in utts.items()))
    r = subprocess.run([sys.executable, ROOT + "/tools/compute_wer.py", "--text", "--mode=strict", "ark:%s" % (tmp_path / "ref.txt"),
                        "ark:%s" % (tmp_path / "hyp.txt")], capture_output=True, text=True)
    assert r.stdout.splitlines()[0].startswith("%WER 0.00 [ 0 / "), r.stdout           # "perfectly recognized"
    # the lattices: device search on the device log-likelihoods == the oracle decoder on the same matrix
    g.tid2pdf = id2pdf
    cfg = abi.decoder_config_recipe(); cfg.beam, cfg.lattice_beam = 13.0, 6.0
    ll = dec_am.loglikes(x0, 0.1)
    d = decoder.LatticeFasterDecoder(decoder.Graph(g), cfg, abi.DecoderSizes(1, 1 << 14, 1 << 18, 1 << 19, 512))
    d.Decode(ll)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    assert lattices_equal(d.GetRawLattice(), o.GetRawLattice()), lattice_diff(d.GetRawLattice(), o.GetRawLattice())
    got = {k: latbin.best_path(l)[0] for k, l in latbin.read_lattices("ark:%s" % (tmp_path / "lat.ark"))}
    assert got == {k: [int(w) for w in ws] for k, (_, ws) in utts.items()}
