import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.')
import bench
from kaldi_amd import abi, batch, synth, decoder, pipeline
from kaldi_amd._lib import lib
sys.argv = ['x', '--utts', '8', '--max-seconds', '2']
args = bench.defaults(bench.parse_args())
g, model, durs, cfg, _ = bench.build_workload(args)
G = decoder.Graph(g)
for peak, noise in [(4.0, 1.5), (5.0, 1.5), (6.0, 1.5), (6.0, 2.0), (8.0, 2.0)]:
    utts = [synth.sample_utterance(g, n_words=6 + i % 7, seed=7000 + i, peak=peak, noise=noise) for i in range(32)]
    T = max(ll.shape[0] for ll, _, _ in utts)
    sz = pipeline.default_sizes(cfg, 32, T + 2, T + 2, hash_capacity=1 << 20, tokens_per_frame=60000)
    bd = decoder.BatchDecoder(G, cfg, sz)
    lats, recs, ms = None, None, None
    try:
        lats, recs, ms = bd.decode_queue([u[0] for u in utts], resident_lanes=32)
    except Exception as e:
        print(peak, noise, "error", e, flush=True)
        continue
    errs = nref = 0
    for i, (ll, words, _) in enumerate(utts):
        bp = decoder.lattice_best_path(lats[i])
        hyp = bp["words"].tolist() if bp else []
        errs += bench._edit_distance(words, hyp); nref += len(words)
    c = np.sum([np.asarray(r.counters[:7], np.float64) for r in recs], axis=0)
    print("peak %.1f noise %.1f: ll std %.2f WER %.1f%% (%d/%d) tokens/frame %.0f expanded %.0f arcs %.0f lattice states/utt %.0f kernel %.1f ms" %
          (peak, noise, float(np.mean([u[0].std(axis=1).mean() for u in utts])), 100.0 * errs / nref, errs, nref, c[5] / c[6], c[0] / c[6], c[1] / c[6],
           np.mean([l.frame.size for l in lats if l is not None]), ms), flush=True)
    del bd
