#!/usr/bin/env python3
"""Work-queue kernel vs the three-launch batch path on the same log-likelihoods: lane-time
utilisation and microseconds per frame at several numbers of resident lanes.

  python tools/queue_bench.py --utts 256 --lanes 64,128,256
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kaldi_amd import abi, decoder, pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mini_librispeech")
    ap.add_argument("--utts", type=int, default=128)
    ap.add_argument("--lanes", default="32,64,128,256")
    ap.add_argument("--graph", default="")
    ap.add_argument("--vocab", type=int, default=0)
    ap.add_argument("--n-hist", type=int, default=0)
    ap.add_argument("--lm-scale", type=float, default=-1.0)
    ap.add_argument("--ll-std", type=float, default=-1.0)
    ap.add_argument("--output-scale", type=float, default=1.0)
    ap.add_argument("--max-seconds", type=float, default=0.0)
    ap.add_argument("--hash-capacity", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    args = bench.defaults(args)
    from kaldi_amd import synth
    g, model, durs, cfg, _ = bench.build_workload(args)
    bench.calibrate(model, args.ll_std)
    waves = synth.make_waves_fast(durs, seed=1000)
    audio = sum(w.size for w in waves) / 16000.0
    max_s = max(w.size for w in waves) / 16000.0 + 0.5
    fps0 = 100.0 / model.subsampling
    psz = pipeline.default_sizes(cfg, len(waves), int(max_s * fps0) + 2, int(audio / len(waves) * fps0) + 2,
                                 hash_capacity=args.hash_capacity or None)
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=len(waves), max_seconds=max_s,
                             avg_seconds=audio / len(waves), sizes=psz)
    pipe.load(waves)
    ms = pipe.run()
    ms = pipe.run()
    adv_batch = pipe.dec.last_advance_ms()
    frames = [int(pipe.dec.counters(u)[6]) for u in range(len(waves))]
    lls = [decoder.DeviceMatrix(pipe.loglikes(u)) for u in range(len(waves))]
    ref = [decoder.get_raw_lattice(pipe.dec._dec, u) for u in range(min(8, len(waves)))]
    out = {"utts": len(waves), "audio_s": audio, "frames": sum(frames), "longest_frames": max(frames),
           "batch_stage_ms": ms, "batch_advance_ms": adv_batch,
           "batch_lane_time_utilisation": sum(frames) / (len(frames) * max(frames)), "queue": []}
    del pipe
    lanes_list = [int(x) for x in args.lanes.split(",")]
    fps = 100.0 / model.subsampling
    sz = pipeline.default_sizes(cfg, max(lanes_list), int(max_s * fps) + 2, int(max_s * fps) + 2,
                                hash_capacity=args.hash_capacity or None)
    bd = decoder.BatchDecoder(decoder.Graph(g), cfg, sz)
    for R in lanes_list:
        best = None
        for _ in range(args.reps):
            t0 = time.time()
            lats, recs, kms = bd.decode_queue(lls, resident_lanes=R)
            wall = time.time() - t0
            best = kms if best is None else min(best, kms)
        from tests.util import lattices_equal
        same = all(lattices_equal(lats[u], ref[u]) for u in range(len(ref)))
        errs = sum(1 for r in recs if r.error)
        out["queue"].append({"lanes": R, "kernel_ms": best, "us_per_frame_per_lane": 1e3 * best * min(R, len(lls)) / sum(frames),
                             "x_rt_decode_only": audio / (best * 1e-3), "same_as_batch": same, "errors": errs,
                             "host_wall_incl_fetch_s": wall})
    print(json.dumps(out, default=float))


if __name__ == "__main__":
    main()
