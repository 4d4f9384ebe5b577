"""How often does the reference's token-order dependence change the result?  Decodes bench
utterances (device log-likelihoods) with the oracle in mode 0 (order-faithful) and mode 1
(canonical) and compares final lattices and best paths.  Run on the GPU box."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np
from kaldi_amd import abi, nnet, pipeline, synth
from oracle import orc
from tests.util import lattices_equal

ap = argparse.ArgumentParser()
ap.add_argument("--ll-std", type=float, default=1.3)
ap.add_argument("--utts", type=int, default=12)
a = ap.parse_args()
g = synth.make_hclg(num_units=1164, vocab=20000, n_hist=18000, fanout=(12, 64), pron_len=(3, 7), seed=2,
                    self_loop_prob=0.5, lm_scale=0.1)
model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs)
cfg = abi.decoder_config_recipe()
durs = np.minimum(synth.utterance_durations(a.utts, seed=1000), 8.0)
waves = [synth.make_wave(d, seed=i) for i, d in enumerate(durs)]
pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=a.utts, max_seconds=8.5)
pipe.load(waves[:1]); pipe.run()
k = a.ll_std / float(np.mean(np.std(pipe.loglikes(0), axis=1)))
out = model.layers[-1]
out.W = (out.W * k).astype(np.float32); out.bias = (out.bias * k).astype(np.float32)
pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=a.utts, max_seconds=8.5)
pipe.decode(waves, lattices=False)
same_lat = same_words = 0
dcost = []
sizes = []
for u in range(a.utts):
    ll = pipe.loglikes(u)
    lats = []
    for mode in (0, 1):
        d = orc.Decoder(g, cfg, mode)
        d.Decode(ll)
        lats.append(d.GetRawLattice())
    b0, b1 = lats[0].best_path(), lats[1].best_path()
    same_lat += lattices_equal(lats[0], lats[1])
    same_words += b0["words"].tolist() == b1["words"].tolist()
    dcost.append(abs((b0["graph_cost"] + b0["acoustic_cost"]) - (b1["graph_cost"] + b1["acoustic_cost"])))
    sizes.append((lats[0].frame.size, lats[1].frame.size, lats[0].arcs.size, lats[1].arcs.size))
print("ll-std %.2f: %d utterances; identical final lattices %d; identical 1-best words %d; max |d cost| %.3g"
      % (a.ll_std, a.utts, same_lat, same_words, max(dcost)))
print("lattice sizes (faithful states, canonical states, faithful arcs, canonical arcs):", sizes[:6])
