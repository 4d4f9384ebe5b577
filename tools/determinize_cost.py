"""Host cost of the lattice tail at bench scale: GetRawLattice (device -> host, canonical
numbering), DeterminizeLatticePhonePruned, CompactLattice write, per utterance, one host thread.
Run on the GPU box."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, nnet, pipeline, synth
from kaldi_amd import io as kio
from bench import calibrate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = synth.make_hclg(num_units=1164, vocab=20000, n_hist=18000, fanout=(12, 64), pron_len=(3, 7), seed=2,
                    self_loop_prob=0.5, lm_scale=0.1)
model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs)
calibrate(model, 1.3)
cfg = abi.decoder_config_recipe()
durs = synth.utterance_durations(n, seed=1000)
waves = [synth.make_wave(d, seed=i) for i, d in enumerate(durs)]
pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=n, max_seconds=float(durs.max()) + 0.5,
                         avg_seconds=float(durs.mean()))
pipe.load(waves)
pipe.run()
tid_phone = np.zeros(g.tid2pdf.size, np.int32)
tid_phone[1::2] = np.arange(1, (g.tid2pdf.size - 1) // 2 + 1)
t_fetch = t_det = t_write = 0.0
states = arcs = dstates = darcs = 0
with tempfile.TemporaryDirectory() as tmp:
    for u in range(n):
        t0 = time.perf_counter()
        lat = decoder.get_raw_lattice(pipe.dec._dec, u)
        t1 = time.perf_counter()
        cl = kio.determinize_lattice(lat, cfg.lattice_beam, tid_phone)
        t2 = time.perf_counter()
        cl.write(os.path.join(tmp, "lat.1"), "utt%d" % u, binary=True, append=u > 0)
        t3 = time.perf_counter()
        t_fetch += t1 - t0; t_det += t2 - t1; t_write += t3 - t2
        states += lat.frame.size; arcs += lat.arcs.size; dstates += cl.num_states; darcs += cl.arcs.size
audio = float(durs.sum())
print("%d utterances, %.0f s audio: raw lattices %d states / %d arcs -> determinized %d states / %d arcs" % (n, audio, states, arcs, dstates, darcs))
print("per utterance on one host thread: fetch+canonicalise %.2f ms, determinize %.2f ms, write %.2f ms  (host tail = %.0f x real time per thread)"
      % (1e3 * t_fetch / n, 1e3 * t_det / n, 1e3 * t_write / n, audio / (t_fetch + t_det + t_write)))
