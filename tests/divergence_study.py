#!/usr/bin/env python3
"""CPU study of the headline load's divergence between the order-free search the device runs (oracle mode 2, which the
device equals bit for bit) and the reference's order-dependent search (oracle mode 0): for a duration-stratified sample of
the planted test set, signed best-path cost gaps, both hypotheses' errors against the planted transcript, and the frames
where the two searches' cutoffs part.  No GPU: the planted log-likelihoods are drawn by numpy with the statistics of
kaldi_amd/csrc/synth.hip (noise * N(0,1) on every pdf, + peak on the path's pdf), not its exact stream.
    python tests/divergence_study.py --utts 138 --threads 8 [--modes 0,2,3]"""
import argparse
import json
import os
import pickle
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=138)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--modes", default="0,2", help="oracle search modes; 0 is the base.  A mode written m@r runs mode m with --hash-ratio r (the reference's "
                                                   "own option, which only changes the HashList's bucket order): 0@3 is the reference against itself")
    ap.add_argument("--peak", type=float, default=8.3)
    ap.add_argument("--noise", type=float, default=3.0)
    ap.add_argument("--cache", default="/tmp/study/graph.pkl")
    ap.add_argument("--max-seconds", type=float, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    sys.argv = ["bench.py"]
    import bench
    from kaldi_amd import synth
    from oracle import orc
    args = bench.defaults(bench.parse_args())
    t0 = time.time()
    if os.path.exists(a.cache):
        g, durs, cfg = pickle.load(open(a.cache, "rb"))
    else:
        g, _, durs, cfg, _ = bench.build_workload(args)
        os.makedirs(os.path.dirname(a.cache), exist_ok=True)
        pickle.dump((g, durs, cfg), open(a.cache, "wb"), protocol=4)
    print("graph %.1f s" % (time.time() - t0), flush=True)
    order = np.argsort(durs)
    k = max(1, len(order) // a.utts)
    sample = [int(u) for u in order[k // 2::k]]
    if a.max_seconds:
        sample = [u for u in sample if durs[u] <= a.max_seconds]
    sample.sort(key=lambda u: -durs[u])
    modes = [m if "@" in m else int(m) for m in a.modes.split(",")]
    res = {}
    lock = threading.Lock()
    todo = list(sample)

    def work():
        while True:
            with lock:
                if not todo:
                    return
                u = todo.pop(0)
            n_words = max(1, int(round(float(durs[u]) * 3.0)))
            words, path = synth.sample_path(g, n_words, seed=900000 + u)
            ll = synth.planted_loglikes_host(path, g.num_pdfs, a.peak, a.noise, seed=5000 + u)
            r = {"dur": float(durs[u]), "ref": [int(w) for w in words], "frames": int(path.size)}
            for m in modes:
                if isinstance(m, str):
                    import copy
                    c2 = copy.copy(cfg)
                    c2.hash_ratio = float(m.split("@")[1])
                    d = orc.Decoder(g, c2, int(m.split("@")[0]))
                else:
                    d = orc.Decoder(g, cfg, m)
                t1 = time.time()
                d.Decode(ll)
                lat = d.GetRawLattice()
                bp = lat.best_path() if lat is not None else None
                nt, cu, of = d.trace()
                r[m] = {"words": [int(w) for w in bp["words"]] if bp is not None else None,
                        "cost": float(bp["graph_cost"] + bp["acoustic_cost"]) if bp is not None else None,
                        "s": time.time() - t1, "nt": nt, "cu": cu}
            with lock:
                res[u] = r
                print("utt %d (%.1f s) done, %d left" % (u, durs[u], len(todo)), flush=True)

    ths = [threading.Thread(target=work) for _ in range(a.threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    base = 0
    rep = {"utterances": len(sample), "peak": a.peak, "noise": a.noise}
    ref_words = sum(len(res[u]["ref"]) for u in sample)
    for m in modes:
        errs = sum(bench._edit_distance(res[u]["ref"], res[u][m]["words"] or []) for u in sample)
        rep["wer_mode%s" % m] = {"errors": errs, "ref_words": ref_words, "wer": 100.0 * errs / ref_words,
                                 "cpu_s": sum(res[u][m]["s"] for u in sample)}
    for m in modes:
        if m == base:
            continue
        gaps = []
        for u in sample:
            c0, cm = res[u][base]["cost"], res[u][m]["cost"]
            if c0 is None or cm is None or abs(cm - c0) > 1e-3 or res[u][base]["words"] != res[u][m]["words"]:
                nt0, ntm = res[u][base]["nt"], res[u][m]["nt"]
                cu0, cum = res[u][base]["cu"], res[u][m]["cu"]
                first = next((int(t) for t in range(min(len(cu0), len(cum))) if cu0[t] != cum[t]), None)
                gaps.append({"utt": u, "dur": res[u]["dur"], "cost_mode%s_minus_mode0" % m: None if c0 is None or cm is None else cm - c0,
                             "errs_mode0": bench._edit_distance(res[u]["ref"], res[u][base]["words"] or []),
                             "errs_mode%s" % m: bench._edit_distance(res[u]["ref"], res[u][m]["words"] or []),
                             "hyp_edit": bench._edit_distance(res[u][base]["words"] or [], res[u][m]["words"] or []),
                             "first_frame_with_other_cutoff": first,
                             "frames_with_tighter_cutoff": int(np.sum(cum[:len(cu0)] < cu0[:len(cum)] - 1e-6)),
                             "frames_with_looser_cutoff": int(np.sum(cum[:len(cu0)] > cu0[:len(cum)] + 1e-6))})
        rep["mode%s_vs_mode0" % m] = {"utterances_differing": len(gaps),
                                      "worse_cost": sum(1 for x in gaps if (x["cost_mode%s_minus_mode0" % m] or 0) > 1e-3),
                                      "better_cost": sum(1 for x in gaps if (x["cost_mode%s_minus_mode0" % m] or 0) < -1e-3),
                                      "abs_wer_delta": abs(rep["wer_mode%s" % m]["wer"] - rep["wer_mode%s" % base]["wer"]),
                                      "detail": gaps}
    s = json.dumps(rep, indent=1, default=float)
    print(s)
    if a.out:
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
