#!/usr/bin/env python3
"""bench.py -- decode RTF (audio-seconds per wall-second) of the MI355X hot path.

One "step" = one pass of the whole hot path over one batch of synthetic utterances that
is already resident in HBM: MFCC -> TDNN-F log-likelihoods -> LatticeFasterDecoder
(init + advance + finalize) -> raw lattices / 1-best staged for the host.
Workload = BASELINE.json configs[1]: mini_librispeech TDNN-F chain topology
(run_tdnn_1h.sh: 768/96, P=2328), tgsmall-scale synthetic HCLG, batch = 64 utterances
per GPU, recipe decoder settings (beam 15, max-active 7000, min-active 200,
lattice-beam 8).  Weak scaling: every rank decodes its own 64-utterance shard.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(c):
    """8 N_exp + 16 A_exp + 8 A_emit + 16 K_surv + 20 L_kept + 12 N_tok."""
    return 8 * c[0] + 16 * c[1] + 8 * c[2] + 16 * c[3] + 20 * c[4] + 12 * c[5]


def build_workload(args, rank):
    from kaldi_amd import abi, nnet, synth
    t0 = time.time()
    if args.workload == "mini_librispeech":
        g = synth.make_hclg(num_units=1164, vocab=args.vocab, n_hist=args.n_hist, fanout=(12, 64),
                            pron_len=(3, 7), seed=2, self_loop_prob=0.5, lm_scale=args.lm_scale)
        model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs, output_scale=args.output_scale,
                                            ivector_dim=100 if args.ivectors else 0)
    elif args.workload == "librispeech":
        g = synth.make_hclg(num_units=3000, vocab=args.vocab, n_hist=args.n_hist, fanout=(12, 64),
                            pron_len=(3, 7), seed=2, self_loop_prob=0.5, lm_scale=args.lm_scale)
        model = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs, output_scale=args.output_scale,
                                       ivector_dim=100 if args.ivectors else 0)
    else:  # tiny (CI / CPU-less smoke of the script itself)
        g = synth.make_hclg(num_units=64, vocab=400, n_hist=60, seed=2)
        model = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=args.output_scale)
    durs = synth.utterance_durations(args.utts, seed=1000 + rank)
    if args.max_seconds:
        durs = np.minimum(durs, args.max_seconds)
    waves = [synth.make_wave(d, seed=rank * 100000 + i) for i, d in enumerate(durs)]
    cfg = abi.decoder_config_recipe()
    return g, model, waves, cfg, time.time() - t0


def calibrate(model, target_std, extractor=None):
    """Random weights give arbitrary output scale; rescale the output layer so that the
    per-frame spread of the log-likelihoods across pdfs is `target_std` nats (chain models
    in the wild: a few nats).  Runs on the GPU (this is workload synthesis, not parity)."""
    from kaldi_amd import abi, decoder, feat, synth
    w = synth.make_wave(3.0, seed=424242)
    f = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(w)
    ivd = model.layers[0].ivector_dim
    iv = None
    if ivd:      # a typical i-vector (the last one of the sample), not zeros: it shifts every output
        iv = extractor.extract_online(f)[-1] if extractor is not None else np.zeros(ivd, np.float32)
    ll = decoder.Nnet(model).Forward(f, ivector=iv)
    spread = float(np.mean(np.std(ll, axis=1)))
    k = target_std / spread
    out = model.layers[-1]
    out.W = (out.W * k).astype(np.float32)
    out.bias = (out.bias * k).astype(np.float32)
    return spread, k


PHASES = ["best", "cutoff", "seed", "expand", "expand_hub", "eps_closure", "compact", "fixup",
          "eps_links", "clear", "fin_sweep", "fin_compact", "flat_setup"]


def phase_share(pipe, waves):
    lane = int(np.argmax([w.size for w in waves]))
    c = pipe.dec.phase_cycles(lane).astype(np.float64)[:len(PHASES)]
    tot = max(c.sum(), 1.0)
    return {k: round(float(v / tot), 3) for k, v in zip(PHASES, c)}


def pmc_traffic(args):
    """HBM bytes per AdvanceKernel launch from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json), only when they were taken on this exact workload."""
    key = "%s/%d/%s/%s" % (args.workload, args.utts, args.ll_std, args.lm_scale)
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                except Exception:
                    continue
                if d.get("workload_key") == key:
                    best = d.get("traffic_bytes_per_launch")
    return best


def _edit_distance(a, b):
    """Levenshtein distance between two word sequences (bin/compute-wer.cc semantics)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def cpu_baseline(g, model, waves, cfg, budget_s, gpu_results=None, cores=0):
    """The CPU oracle (a port of the reference path: faithful decoder mode 0) timed on this host's cores on a bounded
    sample of the same workload: one utterance per thread at a time, like nnet3-latgen-faster-parallel (the oracle is C
    called through ctypes, which releases the interpreter lock)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from kaldi_amd import abi
    from oracle import orc
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores if cores > 0 else avail, avail, 64))

    def one(idx):
        w = waves[idx]
        feats = orc.mfcc(abi.mfcc_opts_hires(), w)
        ll = orc.nnet_forward(model, feats)
        d = orc.Decoder(g, cfg, 0)
        d.Decode(ll)
        lat = d.GetRawLattice()
        return idx, (lat.best_path() if lat is not None else None)

    order = [int(i) for i in np.argsort([w.size for w in waves])]
    t0 = time.time()
    first = one(order[0])                  # the shortest utterance, alone: the single-core rate sizes the sample
    t_first = time.time() - t0
    rate = (waves[order[0]].size / 16000.0) / max(t_first, 1e-6)
    audio_budget = rate * budget_s * cores * 0.8
    sample, audio = [], 0.0
    for idx in order[1:]:                  # shortest first, until about budget_s of wall time on `cores` threads
        a = waves[idx].size / 16000.0
        if sample and audio + a > audio_budget:
            break
        sample.append(idx); audio += a
    t0 = time.time()
    if cores == 1 or len(sample) < 2:
        done = [one(i) for i in sample]
    else:
        with ThreadPoolExecutor(max_workers=cores) as ex:
            done = list(ex.map(one, sorted(sample, key=lambda i: -waves[i].size)))      # longest first: less tail imbalance
    wall = time.time() - t0
    errs = ref_words = 0
    for idx, bp in [first] + done:
        if gpu_results is not None and bp is not None and gpu_results[idx] is not None:
            ref = bp["words"].tolist()                         # the CPU path's 1-best is the "reference transcript"
            errs += _edit_distance(ref, gpu_results[idx]["words"].tolist())
            ref_words += len(ref)
    used = min(cores, max(len(sample), 1))
    out = {"value": audio / max(wall, 1e-9), "unit": "audio-sec/wall-sec", "cores": used, "kind": "port",
           "single_core_value": rate,
           "sample": "%d shortest utterance(s) of the batch after the first (%.1f s audio) on %d thread(s), %.1f s wall; the "
                     "shortest one alone gave the single-core rate (%.1f s CPU): whole path MFCC+nnet+LatticeFasterDecoder"
                     "(order-faithful oracle)+best path" % (len(sample), audio, used, wall, t_first)}
    if gpu_results is not None and ref_words > 0:
        # BASELINE's "WER-equal" clause on synthetic data: word errors of the device 1-best
        # against the CPU path's 1-best on the same utterances
        out["wer_vs_cpu_1best"] = {"errors": errs, "ref_words": ref_words, "wer_percent": 100.0 * errs / ref_words}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mini_librispeech", choices=["mini_librispeech", "librispeech", "tiny"])
    ap.add_argument("--utts", type=int, default=64)
    ap.add_argument("--vocab", type=int, default=20000)
    ap.add_argument("--n-hist", type=int, default=18000)
    ap.add_argument("--output-scale", type=float, default=1.0)
    ap.add_argument("--lm-scale", type=float, default=0.1, help="scale on the synthetic LM costs")
    ap.add_argument("--ll-std", type=float, default=1.3,
                    help="per-frame std (nats) of the synthetic log-likelihoods across pdfs after calibration")
    ap.add_argument("--max-seconds", type=float, default=0.0)
    ap.add_argument("--hash-capacity", type=int, default=0, help="tokens of one frame per lane (power of two); "
                    "0 = derived from max-active")
    ap.add_argument("--overlap", default="", help="output-frame indices (e.g. '48') at which the nnet stage is cut in "
                    "time; each later slice's forward runs while the decoder lanes advance over the slice before it.  Same "
                    "lattices, but measured SLOWER at batch 64 (DESIGN.md section 5), so off by default")
    ap.add_argument("--ivectors", action="store_true", help="variant: the model takes 100-dim online i-vectors, estimated "
                    "on the device from the batch's features (512-Gaussian UBM, period 10) and fed chunk by chunk "
                    "(frames-per-chunk 50) like nnet3-latgen-faster --online-ivectors; no CPU baseline for this variant")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the cpu_baseline leg (0 = every core this process may use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real multi-GPU runs; gloo to "
                    "exercise the N>1 code path with several ranks sharing one GPU")
    ap.add_argument("--device", type=int, default=-1, help="override the HIP device (default LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(args.dist_backend)   # nccl = RCCL; only barriers + two scalar reductions use it

    from kaldi_amd import abi, pipeline
    from kaldi_amd._lib import check, lib, require_gpu
    require_gpu()
    check(lib().kamd_set_device(args.device if args.device >= 0 else local_rank))
    def log(msg):
        if args.verbose and rank == 0:
            print("[bench %.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)
    T0 = time.time()
    g, model, waves, cfg, t_build = build_workload(args, rank)
    ie = None
    if args.ivectors:
        from kaldi_amd import feat, ivector
        sample = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(waves[0][:16000 * 5])
        ie = ivector.IvectorExtractor(ivector.make_synthetic(seed=11, feat_mean=sample.mean(0), feat_std=sample.std(0), max_count=100.0))
        log("i-vector extractor created")
        # random first-layer weights would let the 100 i-vector inputs swamp the 3 x 40 cepstra (a trained
        # model sees them through the LDA-like first affine, roughly unit variance in total): damp those
        # columns so that the synthetic search load stays what --ll-std asks for
        ivs = ie.extract_online(sample)
        l0 = model.layers[0]
        nf = l0.W.shape[1] - l0.ivector_dim
        ratio = np.std(l0.W[:, :nf] @ np.tile(sample[:50], (1, nf // sample.shape[1])).T) / max(np.std(l0.W[:, nf:] @ ivs.T), 1e-9)
        l0.W[:, nf:] *= np.float32(0.3 * ratio)
    spread, k = calibrate(model, args.ll_std, ie)
    log("workload built: %d states %d arcs, %d utts" % (g.num_states, g.num_arcs, len(waves)))
    audio = sum(w.size for w in waves) / 16000.0
    max_s = max(w.size for w in waves) / 16000.0 + 0.5
    sizes = None
    if args.hash_capacity:
        fps = 100.0 / model.subsampling
        sizes = pipeline.default_sizes(cfg, len(waves), int(max_s * fps) + 2, int(audio / len(waves) * fps) + 2,
                                       hash_capacity=args.hash_capacity)
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), model, g, cfg, max_utts=len(waves), max_seconds=max_s,
                             avg_seconds=audio / len(waves), sizes=sizes)
    log("pipeline created")
    if ie is not None:
        pipe.set_ivector_extractor(ie, 50)
    pipe.load(waves)                        # inputs resident in HBM before the timed region
    log("batch loaded")

    def sync_all():
        check(lib().kamd_device_synchronize())
        if dist is not None:
            dist.barrier()

    # one unsliced pass first: the stage split and the GEMM rate without any overlap
    plain_ms = pipe.run()
    plain_flops = lib().kamd_nnet_last_flops(pipe.nnet._h)
    log("unsliced pass: stage ms %s" % plain_ms)
    bounds = [int(x) for x in args.overlap.split(",") if x.strip()]
    pipe.set_overlap(bounds)
    for _ in range(args.warmup):
        ms = pipe.run()
        log("warmup step: stage ms %s" % ms)
    sync_all()
    t0 = time.time()
    stage = np.zeros(4)
    adv_ms, launches = [], 1
    for _ in range(args.steps):
        stage += np.asarray(pipe.run())
        adv_ms.append(pipe.dec.last_advance_ms())
        launches = lib().kamd_decoder_last_advance_launches(pipe.dec._dec)
    sync_all()
    dt = time.time() - t0
    log("timed steps done: %.3f s" % dt)
    if dist is not None:
        import torch
        tdev = "cuda" if args.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        a = torch.tensor([audio], device=tdev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        total_audio = float(a.item())
    else:
        total_audio = audio
    if rank != 0:
        return
    counters = np.sum([pipe.dec.counters(u) for u in range(len(waves))], axis=0)
    # counters are reset at InitDecoding: per step = per batch, spread over `launches` AdvanceKernel
    # launches when the nnet stage is sliced (every lane continues where it stopped)
    alg_bytes = float(algorithmic_bytes(counters)) / launches
    adv = float(np.mean(adv_ms)) / launches
    frames = int(counters[6])
    res = pipe.results(lattices=False)
    log("results fetched")
    out = {
        "metric": "decode RTF (audio-sec/wall-sec)",
        "value": total_audio * args.steps / dt,
        "unit": "audio-sec/wall-sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s TDNN-F chain topology (random init), synthetic %s-scale HCLG "
                               "(%d states, %d arcs), batch=%d utterances/GPU (%.0f s audio), beam 15 "
                               "max-active 7000 min-active 200 lattice-beam 8" %
                               (args.workload, "tglarge" if g.num_states > 2e7 else "tgsmall", g.num_states,
                                g.num_arcs, len(waves), audio),
                   "utterances_per_gpu": len(waves), "loglike_std_nats": args.ll_std},
        "stage_ms": {"features": stage[0] / args.steps, "nnet_before_search": stage[1] / args.steps,
                     "decode_advance": stage[2] / args.steps, "decode_finalize": stage[3] / args.steps},
        "stage_ms_unsliced": {"features": plain_ms[0], "nnet": plain_ms[1], "decode_advance": plain_ms[2],
                              "decode_finalize": plain_ms[3]},
        "nnet_overlap": {"slice_bounds_output_frames": bounds, "advance_launches_per_step": launches},
        "decoder": {"frames": frames, "tokens_per_frame": counters[5] / max(frames, 1),
                    "expanded_per_frame": counters[0] / max(frames, 1),
                    "arcs_per_frame": counters[1] / max(frames, 1),
                    "links_per_frame": counters[4] / max(frames, 1),
                    "words_lane0": int(res[0]["words"].size)},
        "roofline": {"bound": "hbm", "kernel": "kamd::AdvanceKernel",
                     "achieved": alg_bytes / (adv * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (adv * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(args),
                     "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": adv, "launches_per_step": launches},
        "phase_share_longest_lane": phase_share(pipe, waves),
        "nnet_tflops": plain_flops / (plain_ms[1] * 1e-3) / 1e12,
        "setup_s": t_build,
    }
    if args.ivectors:
        out["config"]["workload"] += ", 100-dim online i-vectors estimated on the device"
        out["stage_ms"]["features"] = None
        out["stage_ms"]["features_and_ivectors"] = stage[0] / args.steps
    if not args.no_cpu_baseline and world == 1 and not args.ivectors:
        out["cpu_baseline"] = cpu_baseline(g, model, waves, cfg, args.cpu_budget, res, args.cpu_cores)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out, default=float))


if __name__ == "__main__":
    main()
