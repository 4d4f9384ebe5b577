"""The metric's WER clause end to end: utterances with known transcripts are decoded on the device and by the CPU
oracle (order-faithful mode, the reference's algorithm), both lattice sets go through the same host tail
(determinization, CompactLattice archives) and the same scoring chain (lattice-scale | lattice-add-penalty |
lattice-best-path | compute-wer): identical %WER lines, at an error rate that is not trivially zero."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from kaldi_amd import io as kio
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def score(tmp, name, lats, tid_phone, beam, refs, lmwt, wip):
    ark = str(tmp / (name + ".ark"))
    for i, (k, lat) in enumerate(lats.items()):
        kio.determinize_lattice(lat, beam, tid_phone).write(ark, k, binary=True, append=i > 0)
    py = sys.executable
    rspec = "ark:%s %s/tools/lattice_scale.py --inv-acoustic-scale=%g ark:%s ark:- | %s %s/tools/lattice_add_penalty.py --word-ins-penalty=%g ark:- ark:- |" % (
        py, ROOT, lmwt, ark, py, ROOT, wip)
    hyp = str(tmp / (name + ".hyp"))
    r = subprocess.run([py, ROOT + "/tools/lattice_best_path.py", rspec, "ark,t:" + hyp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([py, ROOT + "/tools/compute_wer.py", "--text", "--mode=present", "ark:" + refs, "ark:" + hyp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout.splitlines(), open(hyp).read()


def test_wer_of_device_decoding_equals_cpu_reference(tmp_path):
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=3)
    cfg = abi.decoder_config_recipe()
    tp = np.zeros(g.tid2pdf.size, np.int32); tp[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)
    G = decoder.Graph(g)
    dev, cpu, refs = {}, {}, []
    sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 1024)
    for i in range(16):
        ll, words, _ = synth.sample_utterance(g, n_words=4 + i % 5, seed=200 + i, peak=2.0)     # noisy: errors do occur (~13 % WER)
        key = "utt%02d" % i
        refs.append("%s %s\n" % (key, " ".join(str(w) for w in words)))
        d = decoder.LatticeFasterDecoder(G, cfg, sz)
        d.Decode(ll)
        dev[key] = d.GetRawLattice()
        o = orc.Decoder(g, cfg, 0)                        # mode 0: the reference's own order-dependent algorithm
        o.Decode(ll)
        cpu[key] = o.GetRawLattice()
    (tmp_path / "ref.txt").write_text("".join(refs))
    wers = []
    for lmwt, wip in ((1.0, 0.0), (3.0, 0.5)):
        got, hyp_d = score(tmp_path, "dev_%g" % lmwt, dev, tp, cfg.lattice_beam, str(tmp_path / "ref.txt"), lmwt, wip)
        want, hyp_c = score(tmp_path, "cpu_%g" % lmwt, cpu, tp, cfg.lattice_beam, str(tmp_path / "ref.txt"), lmwt, wip)
        assert got == want, (got, want)
        assert hyp_d == hyp_c
        assert got[2] == "Scored 16 sentences, 0 not present in hyp."
        wers.append(float(got[0].split()[1]))
    assert 0.0 < wers[0] < 40.0, wers                     # neither trivially perfect nor garbage
