#!/usr/bin/env python3
"""bench.py -- decode RTF (audio-seconds per wall-second) of the MI355X hot path.

Workload (default) = BASELINE.json configs[2]: a LibriSpeech-test-clean-sized synthetic test
set (2620 utterances, 5.4 h, durations lognormal 1-35 s), the LibriSpeech TDNN-F chain
topology (run_tdnn_1d.sh:219-249: 1536/160, 17 layers, P = 6000, random init), a
tglarge-scale synthetic HCLG (31 M states, 69 M arcs), recipe decoder settings (beam 15,
max-active 7000, min-active 200, lattice-beam 8).

One "step" = one pass of the whole hot path over the test set, waveforms already resident in
HBM: MFCC -> TDNN-F log-likelihoods -> LatticeFasterDecoder (work queue over one lane per CU:
init + advance + finalize per utterance) -> pruned raw lattices copied to the host, best path
and lattice determinization on host threads, overlapped with the search.  The step ends when
every utterance's 1-best and determinized lattice are on the host.

--gpus N: N ranks, one per GPU, are spawned by this script itself (or by torchrun: RANK /
LOCAL_RANK / WORLD_SIZE in the environment); the ONE test set is partitioned over the ranks by
longest-processing-time-first (kaldi_amd/shard.py), so scaling is "strong".

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD

PHASES = ["fin_fetch", "cutoff", "seed", "expand", "expand_hub", "eps_closure", "compact", "fixup",
          "eps_links", "clear", "fin_sweep", "fin_compact", "flat_setup", "fin_emit", "fin_eps", "fin_stage"]


def algorithmic_bytes(c):
    """8 N_exp + 16 A_exp + 8 A_emit + 16 K_surv + 20 L_kept + 12 N_tok."""
    return 8 * c[0] + 16 * c[1] + 8 * c[2] + 16 * c[3] + 20 * c[4] + 12 * c[5]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="librispeech", choices=["librispeech", "mini_librispeech", "tiny"],
                    help="librispeech = BASELINE configs[2] (headline); mini_librispeech = configs[1] (use --utts 64)")
    ap.add_argument("--utts", type=int, default=0, help="utterances of the WHOLE test set (0 = 2620 for librispeech, "
                    "64 for mini_librispeech, 12 for tiny)")
    ap.add_argument("--graph", default="", choices=["", "tglarge", "tgsmall"], help="synthetic HCLG scale (default: tglarge "
                    "for librispeech, tgsmall otherwise)")
    ap.add_argument("--vocab", type=int, default=0)
    ap.add_argument("--n-hist", type=int, default=0)
    ap.add_argument("--output-scale", type=float, default=1.0)
    ap.add_argument("--lm-scale", type=float, default=-1.0, help="scale on the synthetic LM costs (default per graph)")
    ap.add_argument("--ll-std", type=float, default=-1.0,
                    help="per-frame std (nats) of the synthetic log-likelihoods across pdfs after calibration")
    ap.add_argument("--max-seconds", type=float, default=0.0)
    ap.add_argument("--lanes", type=int, default=0, help="resident decoder lanes per GPU (0 = one per compute unit)")
    ap.add_argument("--host-threads", type=int, default=0, help="host-tail threads per rank (0 = min(32, cores / ranks))")
    ap.add_argument("--no-determinize", action="store_true")
    ap.add_argument("--hash-capacity", type=int, default=0)
    ap.add_argument("--tokens-per-frame", type=int, default=0, help="arena budget per frame and lane (0 = from max-active / free HBM)")
    ap.add_argument("--nnet-pass-frames", type=int, default=1000000)
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the cpu_baseline leg (0 = min(cores, 32))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-wer", action="store_true")
    ap.add_argument("--wer-utts", type=int, default=64)
    ap.add_argument("--search-mode", type=int, default=2, choices=[1, 2], help="kamd_decoder_set_search_mode: 1 canonical (tight), "
                    "2 canonical-loose (every token the reference's order-dependent pruning can create)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real multi-GPU runs; gloo to "
                    "exercise the N>1 code path with several ranks sharing one GPU")
    ap.add_argument("--device", type=int, default=-1, help="override the HIP device of every rank (default LOCAL_RANK)")
    return ap.parse_args()


# ----------------------------------------------------------------------------- launcher
def launch_ranks(args):
    """--gpus N without a torchrun environment: start N fresh rank processes (this parent never
    touches the GPU), one per GPU, relay rank 0's JSON line, exit with the worst return code."""
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


# ----------------------------------------------------------------------------- workload
def defaults(args):
    if args.utts <= 0:
        args.utts = {"librispeech": 2620, "mini_librispeech": 64, "tiny": 12}[args.workload]
    if not args.graph:
        args.graph = "tglarge" if args.workload == "librispeech" else "tgsmall"
    if args.vocab <= 0:
        args.vocab = 200000 if args.graph == "tglarge" else 20000
    if args.n_hist <= 0:
        args.n_hist = 160000 if args.graph == "tglarge" else 18000
    # search-load knobs: random weights carry no speech information, so the spread of the
    # log-likelihoods and the flatness of the synthetic LM set how many hypotheses survive the beam
    if args.lm_scale < 0:
        args.lm_scale = 0.3 if args.graph == "tglarge" else 0.1
    if args.ll_std < 0:
        args.ll_std = 1.9 if args.graph == "tglarge" else 1.3
    if args.hash_capacity <= 0 and args.graph == "tglarge":
        args.hash_capacity = 1 << 20     # the unigram tree's second level: > 1e5 tokens on the frames after a word boundary
    return args


def build_workload(args):
    from kaldi_amd import abi, nnet, synth
    t0 = time.time()
    if args.workload == "mini_librispeech":
        g = synth.make_hclg(num_units=1164, vocab=args.vocab, n_hist=args.n_hist, fanout=(12, 64),
                            pron_len=(3, 7), seed=2, self_loop_prob=0.5, lm_scale=args.lm_scale)
        model = nnet.tdnnf_mini_librispeech(num_pdfs=g.num_pdfs, output_scale=args.output_scale)
    elif args.workload == "librispeech":
        g = synth.make_hclg(num_units=3000, vocab=args.vocab, n_hist=args.n_hist, fanout=(12, 64),
                            pron_len=(3, 7), seed=2, self_loop_prob=0.5, lm_scale=args.lm_scale)
        model = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs, output_scale=args.output_scale)
    else:  # tiny (CI / smoke of the script itself)
        g = synth.make_hclg(num_units=64, vocab=400, n_hist=60, seed=2)
        model = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=args.output_scale)
    # LibriSpeech test-clean: 2620 utterances, 5.4 h; lognormal(6.2 s, 0.6) clipped to [1, 35] s sums to that
    durs = synth.utterance_durations(args.utts, seed=1, mu=6.2 if args.workload != "tiny" else 1.5)
    if args.max_seconds:
        durs = np.minimum(durs, args.max_seconds)
    cfg = abi.decoder_config_recipe()
    return g, model, durs, cfg, time.time() - t0


def calibrate(model, target_std, extractor=None):
    """Random weights give arbitrary output scale; rescale the output layer so that the
    per-frame spread of the log-likelihoods across pdfs is `target_std` nats (chain models
    in the wild: a few nats).  Runs on the GPU (this is workload synthesis, not parity)."""
    from kaldi_amd import abi, decoder, feat, synth
    w = synth.make_waves_fast([3.0], seed=424242)[0]
    f = feat.Mfcc(abi.mfcc_opts_hires()).ComputeFeatures(w)
    ivd = model.layers[0].ivector_dim
    iv = None
    if ivd:      # a typical i-vector (the last one of the sample), not zeros: it shifts every output (tools/online_latency.py)
        iv = extractor.extract_online(f)[-1] if extractor is not None else np.zeros(ivd, np.float32)
    ll = decoder.Nnet(model).Forward(f, ivector=iv) if ivd else decoder.Nnet(model).Forward(f)
    spread = float(np.mean(np.std(ll, axis=1)))
    k = target_std / spread
    out = model.layers[-1]
    out.W = (out.W * k).astype(np.float32)
    out.bias = (out.bias * k).astype(np.float32)
    return spread, k


def _edit_distance(a, b):
    """Levenshtein distance between two word sequences (bin/compute-wer.cc semantics)."""
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def _run_threads(fn, items, threads):
    """items through fn on `threads` threads pulling from one list (every thread stays busy until the list is
    empty); returns (results in item order, wall seconds, sum of per-item seconds)."""
    import threading
    res = [None] * len(items)
    busy = [0.0] * len(items)
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= len(items):
                return
            t = time.time()
            res[k] = fn(items[k])
            busy[k] = time.time() - t

    t0 = time.time()
    ths = [threading.Thread(target=work) for _ in range(max(1, threads))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return res, time.time() - t0, sum(busy)


def cpu_baseline(g, model, waves, cfg, bd, budget_s, cores):
    """The CPU oracle (a port of the reference path; decoder in its order-faithful mode 0) on this host's cores, on a
    bounded sample of the same test set: `cores` threads, each pulling the next utterance (one LatticeFasterDecoder per
    thread, like nnet3-latgen-faster-parallel; the oracle is C behind ctypes, which releases the interpreter lock), every
    thread busy for the whole measurement.  Second leg: the CPU decoder alone on the DEVICE's log-likelihoods
    (latgen-faster-mapped's job), which is also the 1-best parity check of the sampled utterances."""
    from kaldi_amd import abi
    from oracle import orc
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores if cores > 0 else 32, avail))
    order = [int(i) for i in np.argsort([w.size for w in waves])]

    def whole(idx):
        feats = orc.mfcc(abi.mfcc_opts_hires(), waves[idx])
        ll = orc.nnet_forward(model, feats)
        d = orc.Decoder(g, cfg, 0)
        d.Decode(ll)
        lat = d.GetRawLattice()
        return lat.best_path() if lat is not None else None

    # the shortest utterance alone: the single-core rate sizes the sample
    t0 = time.time()
    whole(order[0])
    t_first = time.time() - t0
    rate = (waves[order[0]].size / 16000.0) / max(t_first, 1e-6)
    audio_budget = rate * budget_s * cores * 0.5            # half of the budget for each leg
    sample, audio = [], 0.0
    for idx in order:                                        # shortest first: >= 4 utterances per thread where the set allows
        a = waves[idx].size / 16000.0
        if len(sample) >= 4 * cores and audio + a > audio_budget:
            break
        sample.append(idx)
        audio += a
        if len(sample) >= len(order):
            break
    # longest of the sample first, so the tail of the run is made of the short ones
    sample.sort(key=lambda i: -waves[i].size)
    ll_cache = {i: bd.loglikes(i) for i in sample}
    res_w, wall_w, busy_w = _run_threads(whole, sample, cores)
    bd_like = ll_cache

    def dec_cached(idx):
        d = orc.Decoder(g, cfg, 0)
        d.Decode(bd_like[idx])
        lat = d.GetRawLattice()
        return lat.best_path() if lat is not None else None

    res_d, wall_d, busy_d = _run_threads(dec_cached, sample, cores)
    errs_w = errs_d = ref_w = ref_d = cost_diff = 0
    for k, idx in enumerate(sample):
        gpu = bd.output(idx)
        gw = gpu["words"].tolist() if gpu is not None else []
        if res_w[k] is not None:
            ref = res_w[k]["words"].tolist()
            errs_w += _edit_distance(ref, gw)
            ref_w += len(ref)
        if res_d[k] is not None:
            ref = res_d[k]["words"].tolist()
            errs_d += _edit_distance(ref, gw)
            ref_d += len(ref)
            if gpu is None or abs((gpu["graph_cost"] + gpu["acoustic_cost"]) - (res_d[k]["graph_cost"] + res_d[k]["acoustic_cost"])) > 1e-3:
                cost_diff += 1
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    used = min(cores, len(sample))
    return {"value": audio / max(wall_w, 1e-9), "unit": "audio-sec/wall-sec", "cores": used, "kind": "port",
            "cpu_model": cpu_model, "cores_available": avail,
            "per_core_value": audio / max(busy_w, 1e-9),
            "thread_busy_fraction": busy_w / max(wall_w * used, 1e-9),
            "sample": "the %d shortest utterances of the test set (%.1f s audio) on %d threads pulling from one list, %.1f s "
                      "wall, %.1f core-s: whole path MFCC + nnet + LatticeFasterDecoder (order-faithful oracle, mode 0) + best "
                      "path; single-utterance probe %.2f s" % (len(sample), audio, used, wall_w, busy_w, t_first),
            "decoder_only": {"value": audio / max(wall_d, 1e-9), "per_core_value": audio / max(busy_d, 1e-9), "cores": used,
                             "wall_s": wall_d,
                             "what": "the CPU decoder alone (mode 0) on the device's log-likelihoods of the same utterances"},
            "one_best_vs_cpu_whole_path": {"errors": errs_w, "ref_words": ref_w},
            "one_best_vs_cpu_decoder_same_loglikes": {"errors": errs_d, "ref_words": ref_d, "utterances_with_other_cost": cost_diff}}


def wer_leg(g, cfg, n_utts, cores, log, hash_capacity=0, search_mode=2):
    """BASELINE's WER clause on synthetic data with a KNOWN transcript: utterances planted in the bench graph
    (synth.sample_utterance: a random word sequence through HCLG, log-likelihoods peaked on the true pdfs at a noise level
    that leaves real errors), decoded by the device (work queue) and by the CPU oracle in its order-faithful mode 0; both
    lattice sets go through determinization and a best path, and are scored against the transcript."""
    from kaldi_amd import decoder, io as kio, latbin, pipeline, synth
    from oracle import orc
    utts = []
    for i in range(n_utts):
        ll, words, _ = synth.sample_utterance(g, n_words=6 + i % 7, seed=7000 + i, peak=3.5, noise=1.5)
        utts.append((ll, words))
    T = max(ll.shape[0] for ll, _ in utts)
    sz = pipeline.default_sizes(cfg, min(n_utts, 64), T + 2, T + 2, hash_capacity=hash_capacity or None, tokens_per_frame=80000)   # flat planted scores: a saturated search
    bd = decoder.BatchDecoder(decoder.Graph(g), cfg, sz)
    bd.SetSearchMode(search_mode)
    lats, recs, ms = bd.decode_queue([ll for ll, _ in utts], resident_lanes=min(n_utts, 64))
    log("wer leg: device decode of %d planted utterances %.1f ms (search mode %d)" % (n_utts, ms, search_mode))
    bd.SetSearchMode(3 - search_mode)
    lats_other, _, ms_o = bd.decode_queue([ll for ll, _ in utts], resident_lanes=min(n_utts, 64))
    log("wer leg: device decode in the other search mode %.1f ms" % ms_o)

    def cpu(i):
        o = orc.Decoder(g, cfg, 0)
        o.Decode(utts[i][0])
        return o.GetRawLattice()

    cpu_lats, wall, _ = _run_threads(cpu, list(range(n_utts)), cores)
    log("wer leg: cpu decode %.1f s" % wall)

    def one_best(lat):
        """determinize, then lattice-best-path's CompactLatticeShortestPath (kaldi_amd/latbin.py)"""
        if lat is None:
            return []
        cl = kio.determinize_lattice(lat, cfg.lattice_beam)
        L = latbin.Lat(cl.start)
        for s_ in range(cl.num_states):
            L.add_state()
            if np.isfinite(cl.final[2 * s_]):
                L.final[s_] = (cl.final[2 * s_], cl.final[2 * s_ + 1], cl.final_string(s_).tolist())
        for k in range(cl.arcs.size):
            a = cl.arcs[k]
            L.arcs[int(a["src"])].append((int(a["dst"]), int(a["label"]), a["graph_cost"], a["acoustic_cost"], cl.arc_string(k).tolist()))
        bp = latbin.best_path(L)
        return [] if bp is None else list(bp[0])

    from kaldi_amd.decoder import lattices_equal
    ref, hyp_d, hyp_c, hyp_o = {}, {}, {}, {}
    e_between = lat_diff = 0
    for i, (ll, words) in enumerate(utts):
        key = "utt%03d" % i
        ref[key] = [str(w) for w in words]
        hyp_d[key] = [str(w) for w in one_best(lats[i])]
        hyp_o[key] = [str(w) for w in one_best(lats_other[i])]
        hyp_c[key] = [str(w) for w in one_best(cpu_lats[i])]
        e_between += _edit_distance(hyp_c[key], hyp_d[key])
        lat_diff += 0 if lattices_equal(lats[i], cpu_lats[i]) else 1
    wd, wc, wo = (latbin.compute_wer(ref, h, "present") for h in (hyp_d, hyp_c, hyp_o))
    return {"utterances": n_utts, "device_search_mode": search_mode, "wer_line_device": wd[0], "wer_line_cpu_reference_port": wc[0],
            "identical_wer_lines": wd == wc, "wer_line_device_search_mode_%d" % (3 - search_mode): wo[0],
            "word_errors_device_vs_cpu_hypotheses": e_between,
            "utterances_whose_raw_lattice_differs_from_mode0": lat_diff,
            "what": "planted transcripts in the bench HCLG, log-likelihoods peaked on the true pdfs (peak 3.5, noise 1.5); device "
                    "(work queue, canonical search) vs CPU oracle mode 0 (the reference's order-dependent search); both through "
                    "DeterminizeLatticePhonePruned + lattice-best-path + compute-wer, scored against the transcript"}


# ----------------------------------------------------------------------------- rank
def main():
    args = defaults(parse_args())
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        launch_ranks(args)          # does not return
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    T0 = time.time()

    def log(msg):
        if args.verbose and rank == 0:
            print("[bench %.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank if args.device < 0 else args.device)
        dist.init_process_group(args.dist_backend)   # nccl = RCCL; only barriers + scalar reductions use it

    from kaldi_amd import abi, batch, shard, synth
    from kaldi_amd._lib import check, lib, require_gpu
    ndev = require_gpu()
    dev = args.device if args.device >= 0 else local_rank
    if dev >= ndev:
        raise SystemExit("rank %d wants device %d but only %d are visible (use --dist-backend gloo --device 0 to share one)" %
                         (rank, dev, ndev))
    check(lib().kamd_set_device(dev))
    g, model, durs, cfg, t_build = build_workload(args)
    spread, k = calibrate(model, args.ll_std)
    log("workload built: %d states %d arcs, %d utts (%.2f h)" % (g.num_states, g.num_arcs, durs.size, durs.sum() / 3600))
    # ONE test set, partitioned over the ranks (steps/nnet3/decode.sh:96,123: split_data + JOB=1:nj)
    mine = shard.lpt_shards(durs, world)[rank]
    waves = synth.make_waves_fast(durs[mine], seed=1000 + rank)
    audio = sum(w.size for w in waves) / 16000.0
    max_s = float(durs.max()) + 0.5
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    host_threads = args.host_threads or max(1, min(32, cores // world))
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, g, cfg, max_seconds=max_s, resident_lanes=args.lanes,
                                host_threads=host_threads, determinize=not args.no_determinize, keep_raw_lattices=False,
                                nnet_pass_frames=args.nnet_pass_frames, hash_capacity=args.hash_capacity or None,
                                tokens_per_frame=args.tokens_per_frame or None, search_mode=args.search_mode,
                                lattice_pool_bytes=max(1 << 30, int(audio * 3.0e5)),
                                long_lanes=16 if world >= 8 else 0)    # small shards: see kamd_batch_decoder_set_long_decoder
    log("batch decoder created (%d host threads)" % host_threads)
    bd.load(waves)                          # inputs resident in HBM before the timed region
    log("shard loaded: %d utterances, %.0f s audio" % (len(waves), audio))

    def sync_all():
        check(lib().kamd_device_synchronize())
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        st = bd.run()
        log("warmup step: feat %.1f nnet %.1f decode %.1f tail %.1f total %.1f ms, failed %d" %
            (st.feat_ms, st.nnet_ms, st.decode_ms, st.host_tail_ms, st.total_ms, st.n_failed))
    sync_all()
    t0 = time.time()
    acc = np.zeros(7)
    for _ in range(args.steps):
        st = bd.run()
        acc += [st.feat_ms, st.nnet_ms, st.decode_ms, st.host_tail_ms, st.total_ms, st.first_result_ms, st.host_thread_ms_sum]
    sync_all()
    dt = time.time() - t0
    my_dt = dt
    log("timed steps done: %.3f s" % dt)
    rank_walls, total_audio = [dt], audio
    if dist is not None:
        import torch
        tdev = "cuda" if args.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=tdev, dtype=torch.float64)
        allw = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allw, t)
        rank_walls = [float(x.item()) for x in allw]
        dt = max(rank_walls)
        a = torch.tensor([audio], device=tdev, dtype=torch.float64)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        total_audio = float(a.item())
    if rank != 0:
        return
    acc /= args.steps
    n = len(waves)
    recs = [bd.record(u) for u in range(n)]
    counters = np.sum([np.asarray(r.counters[:8], np.float64) for r in recs], axis=0)
    frames = int(counters[6])
    alg_bytes = float(algorithmic_bytes(counters))
    dec_ms, nnet_ms = float(acc[2]), float(acc[1])
    longest = int(np.argmax([r.n_frames for r in recs]))
    ph = np.asarray(recs[longest].phase_cycles[:len(PHASES)], np.float64)
    n_failed = sum(1 for r in recs if r.error)
    words = [bd.output(u) for u in range(min(n, 200))]
    mean_words = float(np.mean([len(w["words"]) for w in words if w is not None])) if any(w is not None for w in words) else 0.0
    dec_roof = {"bound": "hbm", "kernel": "kamd::DecodeQueueKernel", "achieved": alg_bytes / (dec_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": alg_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(args, "decode_queue"),
                "traffic_source": "profiles/*_pmc.json of this workload (rocprofv3 --pmc passes), not this run",
                "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": dec_ms, "lanes": int(st.lanes),
                "us_per_frame_per_lane": 1e3 * dec_ms * int(st.lanes) / max(frames, 1)}
    nnet_roof = {"bound": "mfma", "kernel": "kamd::TdnnGemmDmaKernel (all layers of all passes)", "achieved": st.nnet_flops / (nnet_ms * 1e-3) / 1e12,
                 "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": st.nnet_flops / (nnet_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                 "traffic": pmc_traffic(args, "gemm_all_layers"),
                 "traffic_source": "profiles/*_pmc.json of this workload (rocprofv3 --pmc passes), not this run; bytes per step over all GEMM launches",
                 "flops_per_step": st.nnet_flops, "stage_ms": nnet_ms, "passes": int(st.nnet_passes)}
    dominant_is_decoder = dec_ms >= nnet_ms
    out = {
        "metric": "decode RTF (audio-sec/wall-sec)",
        "value": total_audio * args.steps / dt,
        "unit": "audio-sec/wall-sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d synthetic utterances (%.2f h, lognormal 1-35 s) sharded over %d GPU(s), %s TDNN-F chain "
                               "topology (random init, P=%d), synthetic %s-scale HCLG (%d states, %d arcs), beam 15 max-active 7000 "
                               "min-active 200 lattice-beam 8, %s lanes/GPU fed by a device work queue, host tail (lattice read from the page-locked pool, best path, "
                               "%s) on %d threads/GPU inside the timed region" %
                               ("LibriSpeech test-clean sized test set" if args.workload == "librispeech" else args.workload + " set",
                                durs.size, durs.sum() / 3600.0, world, args.workload, g.num_pdfs, args.graph, g.num_states, g.num_arcs,
                                int(st.lanes), "no determinization" if args.no_determinize else "lattice determinization", host_threads),
                   "utterances": int(durs.size), "utterances_rank0": n, "loglike_std_nats": args.ll_std, "lm_scale": args.lm_scale,
                   "baseline_config": "configs[2]" if args.workload == "librispeech" and args.graph == "tglarge" else
                                      ("configs[1]" if args.workload == "mini_librispeech" else "other")},
        "device_only_value": audio / ((acc[0] + acc[1] + acc[2]) * 1e-3),
        "stage_ms": {"features": acc[0], "nnet": acc[1], "decode_queue_kernel": acc[2], "host_tail_after_last_utterance": acc[3],
                     "total_wall": acc[4], "first_result_at": acc[5], "host_tail_cpu_ms_all_threads": acc[6]},
        "rank_wall_s": rank_walls, "rank0_wall_s": my_dt,
        "decoder": {"frames": frames, "tokens_per_frame": counters[5] / max(frames, 1),
                    "expanded_per_frame": counters[0] / max(frames, 1),
                    "arcs_per_frame": counters[1] / max(frames, 1),
                    "links_per_frame": counters[4] / max(frames, 1),
                    "failed_utterances": n_failed, "mean_words_per_utterance_first_200": mean_words},
        "roofline": dec_roof if dominant_is_decoder else nnet_roof,
        "roofline_other_stage": nnet_roof if dominant_is_decoder else dec_roof,
        "phase_share_longest_utterance": {k2: round(float(v / max(ph.sum(), 1.0)), 3) for k2, v in zip(PHASES, ph)},
        "setup_s": t_build,
    }
    out["cpu_baseline"] = None
    if not args.no_cpu_baseline and world == 1:
        log("cpu baseline ...")
        try:
            out["cpu_baseline"] = cpu_baseline(g, model, waves, cfg, bd, args.cpu_budget, args.cpu_cores)
        except Exception as e:                      # noqa: BLE001 - the measured line must still be printed
            out["cpu_baseline"] = {"error": repr(e)}
    if not args.no_wer and world == 1:
        log("wer leg ...")
        del bd
        try:
            out["wer"] = wer_leg(g, cfg, args.wer_utts, min(32, cores), log, args.hash_capacity, args.search_mode)
        except Exception as e:                      # noqa: BLE001
            out["wer"] = {"error": repr(e)}
    print(json.dumps(out, default=float))
    sys.stdout.flush()


def pmc_traffic(args, which):
    """HBM bytes per step of one kernel family ("decode_queue": one DecodeQueueKernel launch; "gemm_all_layers") from the
    committed rocprofv3 PMC passes (profiles/*_pmc.json), only when they were taken on this exact workload."""
    key = "%s/%s/%d/%s/%s" % (args.workload, args.graph, args.utts, args.ll_std, args.lm_scale)
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                except Exception:
                    continue
                if d.get("workload_key") == key and which in d:
                    best = d[which].get("traffic_bytes_per_step")
    return best


if __name__ == "__main__":
    main()
