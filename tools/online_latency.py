"""BASELINE configs[4] (online2-wav-nnet3-latgen-faster style streaming): one stream, audio fed
in chunks, per-chunk latency of features + looped nnet + AdvanceDecoding on the device, and the
cost of a partial result (BestPathEnd + TraceBackBestPath).  Run on the GPU box."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kaldi_amd import abi, decoder, nnet, online, synth

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--chunk", type=float, default=0.24)
ap.add_argument("--ll-std", type=float, default=1.3)
ap.add_argument("--streams", type=int, default=0, help="> 0: that many concurrent streams through kamd_stream_batch")
ap.add_argument("--accept-each", action="store_true", help="with --streams: one AcceptWaveform (one upload) per stream instead of accept_many")
ap.add_argument("--partials", action="store_true", help="with --streams: partial best paths of all streams after every tick (one launch)")
ap.add_argument("--partials-incremental", action="store_true", help="with --partials: kamd_decoder_partial_best_paths_incremental (only the frames whose "
                "best-path token changed since the last tick are walked)")
ap.add_argument("--endpointing", action="store_true", help="with --streams: EndpointDetected for all streams after every tick (one traceback launch)")
ap.add_argument("--ivectors", action="store_true", help="with --streams: the model takes 100-dim online i-vectors, estimated per "
                "stream on the device (512-Gaussian UBM) and fed on DecodableNnetLoopedOnline's chunk schedule (--frames-per-chunk 20)")
ap.add_argument("--silence-weighting", action="store_true", help="with --ivectors: --ivector-silence-weighting.* on (every second "
                "phone counts as silence, weight 0.001, max-state-duration 100): one more traceback launch per tick")
a = ap.parse_args()
g = synth.make_hclg(num_units=1164, vocab=20000, n_hist=18000, fanout=(12, 64), pron_len=(3, 7), seed=2,
                    self_loop_prob=0.5, lm_scale=0.1)
model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs, ivector_dim=100 if a.ivectors else 0)
cfg = abi.decoder_config_recipe()
wave = synth.make_wave(a.seconds, seed=7)
# calibrate the output scale like bench.py does
from bench import calibrate
ie = None
if a.ivectors:
    from kaldi_amd import feat, ivector
    sample = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(wave[:16000 * 5])
    ie = ivector.IvectorExtractor(ivector.make_synthetic(seed=11, feat_mean=sample.mean(0), feat_std=sample.std(0), max_count=100.0))
calibrate(model, a.ll_std, ie)
N, G = decoder.Nnet(model), decoder.Graph(g)
if a.streams > 0:
    S = a.streams
    waves = [synth.make_wave(a.seconds, seed=100 + i) for i in range(S)]
    sb = online.StreamBatch(abi.mfcc_opts_hires(), N, G, cfg, S, max_seconds=a.seconds + 1)
    if ie is not None:
        sb.set_ivector_extractor(ie, 20)
        if a.silence_weighting:
            n_tids = len(g.tid2pdf) - 1
            tid2phone = np.concatenate([[0], np.arange(n_tids) // 2 + 1]).astype(np.int32)
            swc = online.OnlineSilenceWeightingConfig(":".join(str(p) for p in range(1, int(tid2phone.max()) + 1, 2)), 0.001, 100.0)
            sb.set_silence_weighting(swc, tid2phone)
    step = int(a.chunk * 16000)
    ep = online.OnlineEndpointConfig()
    num_tids = len(g.tid2pdf) - 1
    tid2phone = np.concatenate([[0], np.arange(num_tids) // 2 + 1]).astype(np.int32)
    sil_phones = [p for p in range(1, int(tid2phone.max()) + 1) if p % 3 != 0]     # arbitrary: two units of three
    for rep in range(2):
        sb.start(np.arange(S))
        lat, ep_ms, ep_sil, pb_ms = [], [], [], []
        for i in range(0, waves[0].size, step):
            t0 = time.perf_counter()
            if a.accept_each:
                for s_ in range(S):
                    sb.accept(s_, waves[s_][i:i + step], input_finished=i + step >= waves[s_].size)
            else:
                sb.accept_many(np.arange(S), [w[i:i + step] for w in waves], [i + step >= w.size for w in waves])
            t1 = time.perf_counter()
            nd = sb.advance(np.arange(S))
            t2 = time.perf_counter()
            if a.endpointing and nd[0] > 0 and i + step < waves[0].size:
                flags, sil_fr = sb.endpoint_detected(ep, np.arange(S), tid2phone, sil_phones)
                ep_ms.append((time.perf_counter() - t2) * 1e3); ep_sil.append(float(np.mean(sil_fr)))
            if a.partials and nd[0] > 0 and i + step < waves[0].size:
                t3 = time.perf_counter()
                pb = sb.partial_best_paths(np.arange(S), incremental=a.partials_incremental)
                pb_ms.append((time.perf_counter() - t3) * 1e3)
            lat.append((t2 - t1, t1 - t0))
        t0 = time.perf_counter()
        sb.finalize(np.arange(S))
        fin = time.perf_counter() - t0
    adv = np.asarray([x[0] for x in lat]) * 1e3
    up = np.asarray([x[1] for x in lat]) * 1e3
    print("%d streams x %.1f s in %.0f ms chunks: %d ticks, %d frames decoded per stream" % (S, a.seconds, a.chunk * 1e3, adv.size, int(nd[0])))
    print("per tick: upload (%s) %.2f ms + features/nnet/AdvanceDecoding for all streams %.2f ms median (p95 %.2f, max %.2f)"
          % ("one copy per stream" if a.accept_each else "one copy for all", np.median(up), np.median(adv), np.percentile(adv, 95), adv.max()))
    if a.partials and pb_ms:
        t3 = time.perf_counter()
        sb.start(np.arange(2)); sb.accept(0, waves[0][:step * 8]); sb.accept(1, waves[1][:step * 8]); sb.advance([0, 1])
        t3 = time.perf_counter(); sb.partial_best_path(0); one_ms = (time.perf_counter() - t3) * 1e3
        print("partial best paths of all streams: %.2f ms median per tick (p95 %.2f, last tick %.2f; %d words in stream 0); one stream alone, "
              "8 chunks in: %.2f ms" % (np.median(pb_ms), np.percentile(pb_ms, 95), pb_ms[-1], len(pb[0]["words"]), one_ms))
    if a.endpointing and ep_ms:
        print("endpointing for all streams: %.2f ms median per tick (p95 %.2f), mean trailing silence %.1f frames"
              % (np.median(ep_ms), np.percentile(ep_ms, 95), np.mean(ep_sil)))
    print("aggregate %.0f x real time (compute only %.0f x); FinalizeDecoding of all streams %.2f ms"
          % (S * a.seconds * 1e3 / (adv.sum() + up.sum()), S * a.seconds * 1e3 / adv.sum(), fin * 1e3))
    sys.exit(0)
for rep in range(2):                      # first pass warms up allocations / code objects
    d = online.SingleUtteranceNnet3Decoder(abi.mfcc_opts_hires(), N, G, cfg, max_seconds=a.seconds + 1)
    step = int(a.chunk * 16000)
    lat, part = [], []
    for i in range(0, wave.size, step):
        t0 = time.perf_counter()
        d.AcceptWaveform(16000.0, wave[i:i + step])
        if i + step >= wave.size:
            d.InputFinished()
        d.AdvanceDecoding()
        lat.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        d.GetBestPath(end_of_utterance=False)
        part.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    d.FinalizeDecoding()
    bp = d.GetBestPath()
    fin = time.perf_counter() - t0
lat, part = np.asarray(lat) * 1e3, np.asarray(part) * 1e3
print("stream of %.1f s in %.0f ms chunks: %d chunks, %d frames decoded" % (a.seconds, a.chunk * 1e3, lat.size, d.NumFramesDecoded()))
print("per chunk (features + nnet + AdvanceDecoding, incl. host sync): median %.2f ms, p95 %.2f ms, max %.2f ms  => %.0f x real time"
      % (np.median(lat), np.percentile(lat, 95), lat.max(), a.seconds * 1e3 / lat.sum()))
print("partial best path (BestPathEnd + incremental traceback): median %.2f ms, max %.2f ms" % (np.median(part), part.max()))
print("FinalizeDecoding + GetBestPath at the end: %.2f ms" % (fin * 1e3))
