#!/usr/bin/env python3
"""What ONE rank of an N-GPU run of the headline bench does, measured on this one GPU: the rank's LPT shard of the test
set (kaldi_amd/shard.py) through the same NnetBatchDecoder step.  The ranks of a real run are independent and their shards
equal to 0.1 % of audio, so total audio / this step time is what N GPUs would give, minus the barrier.

  python tools/shard_probe.py --worlds 1,2,4,8 [--lanes 0] [--steps 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kaldi_amd import abi, batch, shard, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--utts", type=int, default=0)
    ap.add_argument("--long-lanes", type=int, default=0, help="kamd_batch_decoder_set_long_decoder: lanes of the second decoder object")
    ap.add_argument("--long-lanes-from", type=int, default=0, help="use --long-lanes (default 32) only for worlds >= this (bench.py arms the long-utterance "
                    "decoder from 4 ranks up): one run, every row's efficiency against the SAME N = 1 row")
    ap.add_argument("--hbm-fraction", type=float, default=0.40, help="(with --faithful) share of the free HBM the search arenas take")
    ap.add_argument("--nnet-pass-frames", type=int, default=800000, help="(with --faithful) input frames per pass of the acoustic model")
    ap.add_argument("--tokens-per-frame", type=int, default=0, help="arena budget per frame and lane of both decoder objects (0 = from max-active / free HBM)")
    ap.add_argument("--host", action="store_true", help="the waveforms are uploaded inside run() (bench.py's default contract) instead of resident")
    ap.add_argument("--faithful", action="store_true", help="bench.py's round-4 headline: the i-vector model (chunked, device extractor) on planted transcripts")
    ap.add_argument("--no-override", action="store_true", help="(with --faithful) the search reads the model's own output, not planted rows")
    ap.add_argument("--plain-model", action="store_true", help="(with --faithful) the model without the i-vector input, no extractor; planted rows")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]] + (["--utts", str(a.utts)] if a.utts else [])
    args = bench.defaults(bench.parse_args())
    g, model, durs, cfg, _ = bench.build_workload(args)
    extractor = None
    if a.faithful and not a.plain_model:
        model, extractor = bench.ivector_variant(args, g)
    else:
        bench.calibrate(model, args.ll_std)
    out = []
    base = None
    for world in [int(x) for x in a.worlds.split(",")]:
        long_lanes = a.long_lanes
        if a.long_lanes_from > 0:
            long_lanes = (a.long_lanes or 32) if world >= a.long_lanes_from else 0
        mine = shard.lpt_shards(durs, world)[a.rank % world]
        pset = bench.planted_testset(g, durs, mine, synth) if a.faithful else None
        waves = pset["waves"] if a.faithful else synth.make_waves_fast(durs[mine], seed=1000 + a.rank)
        audio = sum(w.size for w in waves) / 16000.0
        bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=pset["max_seconds"] if a.faithful else float(durs.max()) + 0.5,
                                    resident_lanes=a.lanes, host_threads=16, determinize=True, keep_raw_lattices=False, hash_capacity=args.hash_capacity or None,
                                    search_mode=args.search_mode, lattice_pool_bytes=max(1 << 30, int(audio * 3.0e5)), long_lanes=long_lanes,
                                    tokens_per_frame=a.tokens_per_frame or (11000 if a.faithful else None),
                                    **(dict(nnet_pass_frames=a.nnet_pass_frames, hbm_fraction=a.hbm_fraction) if a.faithful else {}))
        planted = None
        if extractor is not None:
            bd.set_ivector_extractor(extractor, 50)
        (bd.load_host if a.host else bd.load)(waves)
        if a.faithful and not a.no_override:
            planted = synth.planted_loglikes_device(np.concatenate([p for _, p in pset["paths"]]), g.num_pdfs, args.planted_peak, args.planted_noise, seed=5)
            bd.set_loglike_override(planted.ptr(0))
        bd.run()
        t0 = time.time()
        acc = np.zeros(5)
        events = [0]
        for _ in range(a.steps):
            st = bd.run()
            acc += [st.feat_ms, st.nnet_ms, st.decode_ms, st.host_tail_ms, st.total_ms]
            long_utts = st.long_utterances
            if st.n_retried:
                print("step %d: %d utterance(s) took the second chance" % (_, st.n_retried), file=sys.stderr, flush=True)
            if st.n_internal_events:
                events[0] += st.n_internal_events
                print("step %d: %d utterance(s) stopped on an internal consistency check" % (_, st.n_internal_events), file=sys.stderr, flush=True)
            if st.n_failed:
                bad = [(u, bd.record(u).error, bd.record(u).n_frames) for u in range(len(waves)) if bd.record(u).error]
                print("step %d: %d failed utterance(s): %s" % (_, st.n_failed, bad[:8]), file=sys.stderr, flush=True)
        dt = (time.time() - t0) / a.steps
        acc /= a.steps
        total_audio = float(durs.sum()) if not a.faithful else audio * world      # (LPT shards: equal audio per rank to 0.1 %)
        rate = total_audio / dt
        if base is None:
            base = rate / world          # (the first world of the list -- 1 unless asked otherwise -- is every row's base)
        row = {"world": world, "utterances_this_rank": len(waves), "audio_this_rank_s": audio, "longest_s": float(durs[mine].max()),
               "step_ms": 1e3 * dt, "feat_ms": acc[0], "nnet_ms": acc[1], "decode_ms": acc[2], "tail_ms": acc[3],
               "implied_x_real_time_all_ranks": rate, "implied_strong_scaling_efficiency": rate / (base * world), "long_utterances": long_utts,
               "internal_events": events[0]}
        out.append(row)
        print(json.dumps(row), flush=True)
        del bd, planted
    if any(r["internal_events"] for r in out):      # a soak is clean when NO search stopped on an internal check, retried or not
        sys.exit(3)
    return out


if __name__ == "__main__":
    main()
