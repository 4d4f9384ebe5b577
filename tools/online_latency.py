"""BASELINE configs[4] (online2-wav-nnet3-latgen-faster style streaming, src/online2bin/online2-wav-nnet3-latgen-faster.cc:
107,211-285): audio fed in chunks, per-chunk latency of features + looped nnet + AdvanceDecoding on the device, and the
cost of a partial result (BestPathEnd + TraceBackBestPath).  Run on the GPU box, or imported by bench.py's `streaming` leg
(single_stream / many_streams return dicts)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def build(ll_std=1.3, ivectors=False, seconds=12.0):
    """The streaming workload: mini_librispeech-sized TDNN-F, a 20 k-word synthetic HCLG, the output scale calibrated like
    bench.py does.  Returns (graph, model, Nnet, Graph, cfg, extractor)."""
    from kaldi_amd import abi, decoder, nnet, synth
    from bench import calibrate
    g = synth.make_hclg(num_units=1164, vocab=20000, n_hist=18000, fanout=(12, 64), pron_len=(3, 7), seed=2,
                        self_loop_prob=0.5, lm_scale=0.1)
    model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs, ivector_dim=100 if ivectors else 0)
    cfg = abi.decoder_config_recipe()
    ie = None
    if ivectors:
        from kaldi_amd import feat, ivector
        sample = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(synth.make_wave(min(seconds, 5.0), seed=7))
        ie = ivector.IvectorExtractor(ivector.make_synthetic(seed=11, feat_mean=sample.mean(0), feat_std=sample.std(0), max_count=100.0))
    calibrate(model, ll_std, ie)
    return g, model, decoder.Nnet(model), decoder.Graph(g), cfg, ie


def single_stream(N, G, cfg, seconds=12.0, chunk=0.24, reps=2):
    """One SingleUtteranceNnet3Decoder: AcceptWaveform + AdvanceDecoding per chunk, a partial best path after each."""
    from kaldi_amd import abi, online, synth
    wave = synth.make_wave(seconds, seed=7)
    step = int(chunk * 16000)
    for _ in range(reps):                      # the first pass warms up allocations / code objects
        d = online.SingleUtteranceNnet3Decoder(abi.mfcc_opts_hires(), N, G, cfg, max_seconds=seconds + 1)
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
        d.GetBestPath()
        fin = time.perf_counter() - t0
        frames = d.NumFramesDecoded()
        del d
    lat, part = np.asarray(lat) * 1e3, np.asarray(part) * 1e3
    return {"seconds": seconds, "chunk_ms": chunk * 1e3, "chunks": int(lat.size), "frames_decoded": int(frames),
            "ms_per_chunk": float(np.median(lat)), "ms_per_chunk_p95": float(np.percentile(lat, 95)), "ms_per_chunk_max": float(lat.max()),
            "x_rt": float(seconds * 1e3 / lat.sum()), "partial_ms": float(np.median(part)), "partial_ms_max": float(part.max()),
            "finalize_ms": float(fin * 1e3)}


def many_streams(g, N, G, cfg, ie, S, seconds=12.0, chunk=0.24, accept_each=False, partials=True, partials_incremental=True,
                 endpointing=False, silence_weighting=False, reps=2, prune_interval=0):
    """S concurrent streams through kamd_stream_batch: one upload and one features / nnet / AdvanceDecoding tick for all."""
    from kaldi_amd import abi, online, synth
    waves = [synth.make_wave(seconds, seed=100 + i) for i in range(S)]
    sb = online.StreamBatch(abi.mfcc_opts_hires(), N, G, cfg, S, max_seconds=seconds + 1)
    num_tids = len(g.tid2pdf) - 1
    tid2phone = np.concatenate([[0], np.arange(num_tids) // 2 + 1]).astype(np.int32)
    if prune_interval > 0:          # LatticeFasterDecoderConfig::prune_interval: PruneActiveTokens of a stream every that many frames
        sb.set_prune_interval(prune_interval)
    if ie is not None:
        sb.set_ivector_extractor(ie, 20)
        if silence_weighting:
            swc = online.OnlineSilenceWeightingConfig(":".join(str(p) for p in range(1, int(tid2phone.max()) + 1, 2)), 0.001, 100.0)
            sb.set_silence_weighting(swc, tid2phone)
    step = int(chunk * 16000)
    ep = online.OnlineEndpointConfig()
    sil_phones = [p for p in range(1, int(tid2phone.max()) + 1) if p % 3 != 0]     # arbitrary: two units of three
    pb = None
    for _ in range(reps):
        sb.start(np.arange(S))
        lat, ep_ms, ep_sil, pb_ms = [], [], [], []
        for i in range(0, waves[0].size, step):
            t0 = time.perf_counter()
            if accept_each:
                for s_ in range(S):
                    sb.accept(s_, waves[s_][i:i + step], input_finished=i + step >= waves[s_].size)
            else:
                sb.accept_many(np.arange(S), [w[i:i + step] for w in waves], [i + step >= w.size for w in waves])
            t1 = time.perf_counter()
            nd = sb.advance(np.arange(S))
            t2 = time.perf_counter()
            if endpointing and nd[0] > 0 and i + step < waves[0].size:
                _, sil_fr = sb.endpoint_detected(ep, np.arange(S), tid2phone, sil_phones)
                ep_ms.append((time.perf_counter() - t2) * 1e3); ep_sil.append(float(np.mean(sil_fr)))
            if partials and nd[0] > 0 and i + step < waves[0].size:
                t3 = time.perf_counter()
                pb = sb.partial_best_paths(np.arange(S), incremental=partials_incremental)
                pb_ms.append((time.perf_counter() - t3) * 1e3)
            lat.append((t2 - t1, t1 - t0))
        cnt = sb.dec.counters(0)                  # stream 0's work counters before the finalize: [N_exp, A_exp, A_emit, K_surv, L_kept, N_tok, frames, ..]
        t0 = time.perf_counter()
        sb.finalize(np.arange(S))
        fin = time.perf_counter() - t0
        t0 = time.perf_counter()
        sb.best_path(0)
        fetch0 = time.perf_counter() - t0
    adv = np.asarray([x[0] for x in lat]) * 1e3
    up = np.asarray([x[1] for x in lat]) * 1e3
    out = {"streams": int(S), "seconds": seconds, "chunk_ms": chunk * 1e3, "ticks": int(adv.size), "frames_decoded_per_stream": int(nd[0]),
           "upload_ms_per_tick": float(np.median(up)), "ms_per_tick": float(np.median(adv)), "ms_per_tick_p95": float(np.percentile(adv, 95)),
           "ms_per_tick_max": float(adv.max()), "aggregate_x_rt": float(S * seconds * 1e3 / (adv.sum() + up.sum())),
           "aggregate_x_rt_compute_only": float(S * seconds * 1e3 / adv.sum()), "finalize_ms": float(fin * 1e3),
           "finalize_ms_per_stream": float(fin * 1e3 / S), "prune_interval": int(prune_interval), "compactions": int(sb.num_compactions()), "lattice_fetch_and_best_path_ms_stream_0": float(fetch0 * 1e3),
           "tokens_per_frame_stream_0": float(cnt[5] / max(cnt[6], 1)), "expanded_per_frame_stream_0": float(cnt[0] / max(cnt[6], 1)),
           "online_ivectors": ie is not None, "silence_weighting": bool(silence_weighting and ie is not None)}
    if partials and pb_ms:
        out.update({"partials_ms_per_tick": float(np.median(pb_ms)), "partials_ms_per_tick_p95": float(np.percentile(pb_ms, 95)),
                    "partials_incremental": bool(partials_incremental), "words_in_stream_0": int(len(pb[0]["words"]))})
    if endpointing and ep_ms:
        out.update({"endpointing_ms_per_tick": float(np.median(ep_ms)), "mean_trailing_silence_frames": float(np.mean(ep_sil))})
    del sb
    return out


def main():
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
    ap.add_argument("--prune-interval", type=int, default=0, help="with --streams: LatticeFasterDecoderConfig::prune_interval -- a stream is pruned "
                    "(kamd_decoder_compact) every that many decoded frames, between ticks (the reference's default: 25); 0 = only at the end")
    ap.add_argument("--silence-weighting", action="store_true", help="with --ivectors: --ivector-silence-weighting.* on (every second "
                    "phone counts as silence, weight 0.001, max-state-duration 100): one more traceback launch per tick")
    a = ap.parse_args()
    g, model, N, G, cfg, ie = build(a.ll_std, a.ivectors, a.seconds)
    if a.streams > 0:
        r = many_streams(g, N, G, cfg, ie, a.streams, a.seconds, a.chunk, a.accept_each, a.partials, a.partials_incremental,
                         a.endpointing, a.silence_weighting, prune_interval=a.prune_interval)
        print("%d streams x %.1f s in %.0f ms chunks: %d ticks, %d frames decoded per stream" % (r["streams"], a.seconds, a.chunk * 1e3, r["ticks"], r["frames_decoded_per_stream"]))
        print("per tick: upload (%s) %.2f ms + features/nnet/AdvanceDecoding for all streams %.2f ms median (p95 %.2f, max %.2f)"
              % ("one copy per stream" if a.accept_each else "one copy for all", r["upload_ms_per_tick"], r["ms_per_tick"], r["ms_per_tick_p95"], r["ms_per_tick_max"]))
        if "partials_ms_per_tick" in r:
            print("partial best paths of all streams: %.2f ms median per tick (p95 %.2f; %d words in stream 0)"
                  % (r["partials_ms_per_tick"], r["partials_ms_per_tick_p95"], r["words_in_stream_0"]))
        if "endpointing_ms_per_tick" in r:
            print("endpointing for all streams: %.2f ms median per tick, mean trailing silence %.1f frames" % (r["endpointing_ms_per_tick"], r["mean_trailing_silence_frames"]))
        print("aggregate %.0f x real time (compute only %.0f x); FinalizeDecoding of all streams %.2f ms (%.3f per stream); stream 0: lattice fetch + "
              "best path %.2f ms, %.0f tokens created / %.0f expanded per frame"
              % (r["aggregate_x_rt"], r["aggregate_x_rt_compute_only"], r["finalize_ms"], r["finalize_ms_per_stream"],
                 r["lattice_fetch_and_best_path_ms_stream_0"], r["tokens_per_frame_stream_0"], r["expanded_per_frame_stream_0"]))
        return
    r = single_stream(N, G, cfg, a.seconds, a.chunk)
    print("stream of %.1f s in %.0f ms chunks: %d chunks, %d frames decoded" % (a.seconds, a.chunk * 1e3, r["chunks"], r["frames_decoded"]))
    print("per chunk (features + nnet + AdvanceDecoding, incl. host sync): median %.2f ms, p95 %.2f ms, max %.2f ms  => %.0f x real time"
          % (r["ms_per_chunk"], r["ms_per_chunk_p95"], r["ms_per_chunk_max"], r["x_rt"]))
    print("partial best path (BestPathEnd + incremental traceback): median %.2f ms, max %.2f ms" % (r["partial_ms"], r["partial_ms_max"]))
    print("FinalizeDecoding + GetBestPath at the end: %.2f ms" % r["finalize_ms"])


if __name__ == "__main__":
    main()
