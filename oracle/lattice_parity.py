"""oracle/lattice_parity.py -- TEST INFRASTRUCTURE ONLY (never shipped): lattice-level comparison of two sets of raw
lattices of the same utterances -- the device's and the CPU oracle's order-faithful (mode 0) ones -- beyond the 1-best:

  * 1-best after DeterminizeLatticePhonePruned (lattice-best-path), scored against the transcript (compute-wer);
  * the 10 best word sequences of each determinized lattice (lattice-to-nbest): how many of the CPU's are in the device's,
    how many utterances have identical lists (words and costs);
  * lattice-oracle: the smallest edit distance between the transcript and any path of the lattice, i.e. what a perfect
    rescoring pass could still recover from it;
  * 1-best after lattice-lmrescore-const-arpa with a SECOND language model (the recipes' published tglarge / fglarge rows
    are rescored tgsmall lattices: egs/librispeech/s5/local/chain/tuning/run_tdnn_1d.sh:314-325,
    latbin/lattice-lmrescore-const-arpa.cc:76-110).

Used by bench.py's wer leg and tests/test_gpu_wer.py; everything here is host Python over kaldi_amd/latbin.py and
kaldi_amd/constarpa.py (the product's own host code for these tools)."""
import os
import tempfile

import numpy as np

from kaldi_amd import constarpa, io as kio, latbin


def second_lm(vocab, n_bigrams=200000, seed=99):
    """A synthetic second LM over word ids 1..vocab as a ConstArpaLm: every unigram (random log-probabilities with
    back-off weights) and `n_bigrams` random bigrams, integer symbols, <s> = vocab + 1, </s> = vocab + 2."""
    rng = np.random.default_rng(seed)
    bos, eos = vocab + 1, vocab + 2
    uni = -np.abs(rng.normal(4.0, 1.0, vocab + 2)).astype(np.float64)          # log10 probabilities
    bo = -np.abs(rng.normal(0.3, 0.1, vocab + 2))
    a = rng.integers(1, vocab + 1, n_bigrams)
    b = rng.integers(1, vocab + 1, n_bigrams)
    key = np.unique(a.astype(np.int64) * (vocab + 3) + b)
    a, b = key // (vocab + 3), key % (vocab + 3)
    big = -np.abs(rng.normal(1.5, 0.7, a.size))
    fd, path = tempfile.mkstemp(suffix=".arpa")
    with os.fdopen(fd, "w") as f:
        f.write("\\data\\\nngram 1=%d\nngram 2=%d\n\n\\1-grams:\n" % (vocab + 2, a.size))
        f.write("-99\t%d\t%.4f\n" % (bos, bo[vocab]))
        f.write("%.4f\t%d\n" % (uni[vocab + 1], eos))
        f.write("".join("%.4f\t%d\t%.4f\n" % (uni[w - 1], w, bo[w - 1]) for w in range(1, vocab + 1)))
        f.write("\n\\2-grams:\n")
        f.write("".join("%.4f\t%d %d\n" % (p, x, y) for p, x, y in zip(big, a, b)))
        f.write("\n\\end\\\n")
    try:
        return constarpa.ConstArpaLm.build(path, bos, eos)
    finally:
        os.unlink(path)


def to_lat(cl):
    """kaldi_amd.io.CompactLattice -> kaldi_amd.latbin.Lat"""
    L = latbin.Lat(cl.start)
    for s in range(cl.num_states):
        L.add_state()
        if np.isfinite(cl.final[2 * s]):
            L.final[s] = (cl.final[2 * s], cl.final[2 * s + 1], cl.final_string(s).tolist())
    for k in range(cl.arcs.size):
        a = cl.arcs[k]
        L.arcs[int(a["src"])].append((int(a["dst"]), int(a["label"]), a["graph_cost"], a["acoustic_cost"], cl.arc_string(k).tolist()))
    return L


def compare(transcripts, device_lats, cpu_lats, lattice_beam, lm=None, lm_scale=1.0, nbest=10):
    """transcripts: [[word ids]]; device_lats / cpu_lats: raw lattices (kaldi_amd.decoder.Lattice or None) per utterance."""
    ref, hyp = {}, {"device": {}, "cpu": {}}
    resc = {"device": {}, "cpu": {}}
    oracle = {"device": 0, "cpu": 0}
    overlap, same_list, n_lists, depth = 0.0, 0, 0, {"device": [], "cpu": []}
    resc_same = resc_n = 0
    for i, words in enumerate(transcripts):
        key = "utt%03d" % i
        ref[key] = [str(w) for w in words]
        L = {}
        for side, lat in (("device", device_lats[i]), ("cpu", cpu_lats[i])):
            if lat is None:
                L[side] = None
                hyp[side][key] = []
                oracle[side] += len(words)
                continue
            cl = kio.determinize_lattice(lat, lattice_beam)
            L[side] = to_lat(cl)
            bp = latbin.best_path(L[side])
            hyp[side][key] = [] if bp is None else [str(w) for w in bp[0]]
            oracle[side] += latbin.oracle_errors(L[side], list(words))
            fin = np.isfinite(cl.final[0::2])
            depth[side].append(float(cl.arcs["str_len"].sum() + cl.final_str_len[fin].sum()) / max(lat.num_frames, 1))
        if L["device"] is not None and L["cpu"] is not None:
            nd, nc = latbin.nbest(L["device"], nbest), latbin.nbest(L["cpu"], nbest)
            sd = {tuple(w) for w, _ in nd}
            overlap += sum(1 for w, _ in nc if tuple(w) in sd) / max(len(nc), 1)
            same_list += int(len(nd) == len(nc) and all(a[0] == b[0] and abs(a[1] - b[1]) <= 1e-3 * max(1.0, abs(b[1])) for a, b in zip(nd, nc)))
            n_lists += 1
            if lm is not None:
                out = {}
                for side in ("device", "cpu"):
                    R = lm.rescore(L[side], lm_scale)
                    bp = None if R is None else latbin.best_path(R)
                    out[side] = [] if bp is None else [str(w) for w in bp[0]]
                    resc[side][key] = out[side]
                resc_same += int(out["device"] == out["cpu"])
                resc_n += 1
    nref = sum(len(r) for r in ref.values())
    res = {"utterances": len(transcripts), "reference_words": nref,
           "wer_line_device": latbin.compute_wer(ref, hyp["device"], "present")[0],
           "wer_line_cpu_mode0": latbin.compute_wer(ref, hyp["cpu"], "present")[0],
           "one_best_identical_utterances": sum(1 for k in ref if hyp["device"][k] == hyp["cpu"][k]),
           "nbest": nbest, "nbest_overlap": overlap / max(n_lists, 1), "nbest_identical_lists": same_list, "nbest_compared": n_lists,
           "lattice_oracle_wer_device": 100.0 * oracle["device"] / max(nref, 1), "lattice_oracle_wer_cpu_mode0": 100.0 * oracle["cpu"] / max(nref, 1),
           "lattice_oracle_errors_device": oracle["device"], "lattice_oracle_errors_cpu_mode0": oracle["cpu"],
           "lattice_depth_device": float(np.mean(depth["device"])) if depth["device"] else None,
           "lattice_depth_cpu_mode0": float(np.mean(depth["cpu"])) if depth["cpu"] else None}
    if lm is not None:
        res.update({"rescored_wer_line_device": latbin.compute_wer(ref, resc["device"], "present")[0],
                    "rescored_wer_line_cpu_mode0": latbin.compute_wer(ref, resc["cpu"], "present")[0],
                    "rescored_one_best_identical_utterances": resc_same, "rescored_compared": resc_n, "rescoring_lm_scale": lm_scale})
    return res
