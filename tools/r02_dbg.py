"""Search-load sweep on the bench workload: tokens / arcs per frame and time per frame for (lm-scale, ll-std) pairs."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import bench
from kaldi_amd import abi, batch, synth, decoder
pairs = [tuple(float(x) for x in p.split(',')) for p in sys.argv[1].split(';')]
rest = sys.argv[2:]
last_lm, g = None, None
for lm, std in pairs:
    sys.argv = ['x', '--utts', '16', '--max-seconds', '4', '--lm-scale', str(lm), '--ll-std', str(std)] + rest
    args = bench.defaults(bench.parse_args())
    if lm != last_lm:
        g, model0, durs, cfg, _ = bench.build_workload(args)
        G = decoder.Graph(g)
        last_lm = lm
    from kaldi_amd import nnet
    model = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs) if args.workload == 'librispeech' else nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs)
    bench.calibrate(model, args.ll_std)
    waves = synth.make_waves_fast(durs, seed=1000)
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, G, cfg, max_seconds=5.0, resident_lanes=16, host_threads=4,
                                hash_capacity=args.hash_capacity or (1 << 20))
    bd.load(waves)
    bd.run()
    st = bd.run()
    c = np.sum([np.asarray(bd.record(u).counters[:7], np.float64) for u in range(len(waves))], axis=0)
    errs = [bd.record(u).error for u in range(len(waves))]
    fr = max(c[6], 1)
    mx = max(bd.record(u).n_frames for u in range(len(waves)))
    print("lm %.2f std %.2f: failed %d, frames %d, expanded/frame %.0f arcs/frame %.0f tokens/frame %.0f links/frame %.0f, kernel %.1f ms = %.0f us/frame (longest lane %d frames)"
          % (lm, std, sum(1 for e in errs if e), c[6], c[0] / fr, c[1] / fr, c[5] / fr, c[4] / fr, st.decode_ms, 1e3 * st.decode_ms / mx, mx), flush=True)
    del bd
