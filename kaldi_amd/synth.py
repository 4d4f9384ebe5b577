"""Seeded synthetic workloads (no corpora / models / HCLG files exist offline).

* `make_hclg`  -- an HCLG-like decoding graph in OpenFst arc order (CSR of
  fst::StdArc records): LM history states (hubs) with word arcs + epsilon back-off
  arcs, per-(history, word) chains of HMM states (chain topology: one state per
  phone-unit with a self-loop tid and a forward tid, 2 pdfs per unit), epsilon arcs
  from word ends back into the LM, finals on LM states.  First-order statistics follow
  SURVEY.md 8(d): ~2-3 arcs/state, ~10-15 % epsilon arcs, hub states with 10^2-10^4
  arcs, olabels on ~1/8 of arcs, no epsilon cycles.
* `sample_utterance` -- a random path through that graph (truth transcript) and
  log-likelihood matrices peaked on the true pdf (a well-conditioned decode).
* `make_wave` -- band-limited noise + sinusoids at int16 scale (Kaldi wave scale).
"""
from dataclasses import dataclass

import numpy as np

from .abi import ARC_DTYPE


@dataclass
class Hclg:
    num_states: int
    start: int
    arc_off: np.ndarray      # int64 [S+1]
    arcs: np.ndarray         # ARC_DTYPE [A]
    final: np.ndarray        # float32 [S] (+inf = not final)
    tid2pdf: np.ndarray      # int32 [num_tids+1], index 0 unused (=-1)
    num_pdfs: int
    # generator metadata (for sample_utterance)
    n_hist: int = 0
    pair_hist: np.ndarray = None
    pair_word: np.ndarray = None
    pair_base: np.ndarray = None   # [K, Lmax] trie node of each pronunciation position
    pair_len: np.ndarray = None
    hist_pair_off: np.ndarray = None
    word_next_hist: np.ndarray = None
    chain_unit: np.ndarray = None   # unit of each chain state (index = state - n_hist)

    @property
    def num_arcs(self):
        return int(self.arc_off[-1])


def _rank_within(keys):
    """rank of each element among equal keys (stable), and the per-key counts."""
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    first = np.concatenate([[True], sk[1:] != sk[:-1]])
    start = np.maximum.accumulate(np.where(first, np.arange(sk.size), 0))
    rank = np.empty(keys.size, np.int64)
    rank[order] = np.arange(sk.size) - start
    return rank


def make_hclg(num_units=64, vocab=200, n_hist=50, fanout=(4, 24), pron_len=(2, 6),
              seed=2, self_loop_prob=None, lm_scale=1.0):
    """Build the synthetic graph.  States 0..n_hist-1 are LM history states (0 = unigram
    back-off hub, start = min(1, n_hist-1)); the rest are nodes of per-history lexicon
    prefix trees (det(L o G)-like: words sharing a pronunciation prefix share states, so
    a hub's out-degree is bounded by the number of first units, not by the vocabulary).
    LM costs are pushed toward the front: the arc into a trie node costs
    -log(P(node)/P(parent)); the word-end epsilon arc carries the residual and the word."""
    rng = np.random.default_rng(seed)
    U, V, H = num_units, vocab, n_hist
    wlen = rng.integers(pron_len[0], pron_len[1] + 1, V)
    Lmax = int(wlen.max())
    wpron = np.full((V, Lmax), -1, np.int64)
    for d in range(Lmax):
        m = wlen > d
        wpron[m, d] = rng.integers(0, U, int(m.sum()))
    word_next_hist = rng.integers(0, H, V)
    # (history, word) pairs: unigram hub has every word, others a zipf-ish random subset
    hs = [np.zeros(V, np.int64)]
    ws = [np.arange(V, dtype=np.int64)]
    if H > 1:
        fo = rng.integers(fanout[0], fanout[1] + 1, H - 1)
        hh = np.repeat(np.arange(1, H, dtype=np.int64), fo)
        ww = np.minimum((V * rng.random(hh.size) ** 2).astype(np.int64), V - 1)
        key = np.unique(hh * V + ww)
        hs.append(key // V)
        ws.append(key % V)
    pair_hist = np.concatenate(hs)
    pair_word = np.concatenate(ws)
    K = pair_hist.size
    pair_len = wlen[pair_word]
    raw = rng.gamma(1.0, 1.0, K) + 1e-3
    tot = np.bincount(pair_hist, weights=raw, minlength=H)
    pair_p = raw / tot[pair_hist]                       # p(w | h)
    hist_cnt = np.bincount(pair_hist, minlength=H)
    hist_pair_off = np.concatenate([[0], np.cumsum(hist_cnt)])
    # ---- prefix-tree nodes, depth by depth
    pron = wpron[pair_word]                              # [K, Lmax]
    pair_nodes = np.full((K, Lmax), -1, np.int64)
    node_unit, node_parent, node_prob = [], [], []       # parent < 0 encodes LM state -(h+1)
    n_nodes = 0
    parent_id = -(pair_hist + 1)
    for d in range(Lmax):
        valid = pair_len > d
        keyd = (parent_id[valid] + H + 1) * U + pron[valid, d]   # parent ids shifted to >= 0
        uq, inv = np.unique(keyd, return_inverse=True)
        ids = n_nodes + inv
        pair_nodes[valid, d] = ids
        node_unit.append(uq % U)
        node_parent.append(uq // U - (H + 1))
        node_prob.append(np.bincount(inv, weights=pair_p[valid], minlength=uq.size))
        n_nodes += uq.size
        nxt = parent_id.copy()
        nxt[valid] = ids
        parent_id = nxt
    node_unit = np.concatenate(node_unit)
    node_parent = np.concatenate(node_parent)
    node_prob = np.concatenate(node_prob)
    N = n_nodes
    S = H + N
    node_state = H + np.arange(N)
    par_is_lm = node_parent < 0
    par_state = np.where(par_is_lm, -(node_parent + 1), H + node_parent)
    par_prob = np.where(par_is_lm, 1.0, node_prob[np.maximum(node_parent, 0)])
    child_cost = -lm_scale * np.log(node_prob / par_prob)   # lm_scale < 1 flattens the LM (wider search)
    # word ends: pair k ends at node pair_nodes[k, len-1]
    end_node = pair_nodes[np.arange(K), pair_len - 1]
    end_cost = -lm_scale * np.log(np.minimum(1.0, pair_p / node_prob[end_node]))
    # ---- arc counts per state
    n_child = np.bincount(par_state, minlength=S)
    n_end = np.bincount(H + end_node, minlength=S)
    narcs = n_child + n_end
    narcs[:H] += (np.arange(H) > 0)          # back-off epsilon
    narcs[H:] += 1                           # self-loop
    arc_off = np.concatenate([[0], np.cumsum(narcs)]).astype(np.int64)
    A = int(arc_off[-1])
    arcs = np.zeros(A, ARC_DTYPE)
    # chain topology: transition probabilities are a fixed 0.5/0.5 (self_loop_prob=0.5);
    # None draws per-state probabilities (more graph-side discrimination, easier search)
    selfp = rng.uniform(0.25, 0.6, N) if self_loop_prob is None else np.full(N, float(self_loop_prob))
    # back-off arcs (first arc of LM states h > 0)
    if H > 1:
        bo = arc_off[1:H]
        arcs["weight"][bo] = lm_scale * rng.uniform(0.5, 3.0, H - 1)
        arcs["nextstate"][bo] = 0
    # self-loops (first arc of every node)
    a0 = arc_off[node_state]
    arcs["ilabel"][a0] = 2 + 2 * node_unit
    arcs["weight"][a0] = -np.log(selfp)
    arcs["nextstate"][a0] = node_state
    # child arcs: after the back-off / self-loop
    lead = np.where(par_state < H, (par_state > 0).astype(np.int64), 1)
    ca = arc_off[par_state] + lead + _rank_within(par_state)
    stay = np.where(par_is_lm, 0.0, -np.log(1.0 - selfp[np.maximum(node_parent, 0)]))
    arcs["ilabel"][ca] = 1 + 2 * node_unit               # forward tid of the child's unit
    arcs["weight"][ca] = child_cost + stay
    arcs["nextstate"][ca] = node_state
    # word-end epsilon arcs: after self-loop and children
    es = H + end_node
    ea = arc_off[es] + 1 + n_child[es] + _rank_within(es)
    arcs["ilabel"][ea] = 0
    arcs["olabel"][ea] = pair_word + 1
    arcs["weight"][ea] = end_cost - np.log(1.0 - selfp[end_node])
    arcs["nextstate"][ea] = word_next_hist[pair_word]
    final = np.full(S, np.inf, np.float32)
    final[:H] = rng.uniform(0.5, 3.0, H)
    tid2pdf = np.full(2 * U + 1, -1, np.int32)
    tid2pdf[1:] = np.arange(2 * U)
    return Hclg(S, min(1, H - 1), arc_off, arcs, final, tid2pdf, 2 * U, n_hist=H,
                pair_hist=pair_hist, pair_word=pair_word, pair_base=pair_nodes,
                pair_len=pair_len, hist_pair_off=hist_pair_off,
                word_next_hist=word_next_hist, chain_unit=node_unit)


def make_random_graph(num_states=300, num_labels=40, mean_arcs=3.0, eps_frac=0.12,
                      final_frac=0.05, seed=1):
    """Unstructured random graph (SURVEY 6 probe style): stresses hashing and the
    epsilon closure.  Epsilon arcs only go to higher state ids (no epsilon cycles)."""
    rng = np.random.default_rng(seed)
    S = num_states
    n = np.maximum(1, rng.poisson(mean_arcs, S))
    arc_off = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    A = int(arc_off[-1])
    src = np.repeat(np.arange(S), n)
    arcs = np.zeros(A, ARC_DTYPE)
    is_eps = (rng.random(A) < eps_frac) & (src < S - 1)
    arcs["ilabel"] = np.where(is_eps, 0, rng.integers(1, num_labels + 1, A))
    arcs["olabel"] = np.where(rng.random(A) < 0.125, rng.integers(1, 50, A), 0)
    arcs["weight"] = rng.uniform(0.05, 3.0, A)
    nxt = rng.integers(0, S, A)
    nxt_eps = src + 1 + (rng.random(A) * (S - 1 - src)).astype(np.int64)
    arcs["nextstate"] = np.where(is_eps, np.minimum(nxt_eps, S - 1), nxt)
    final = np.where(rng.random(S) < final_frac, rng.uniform(0, 2, S), np.inf).astype(np.float32)
    tid2pdf = np.full(num_labels + 1, -1, np.int32)
    tid2pdf[1:] = rng.integers(0, max(2, num_labels // 2), num_labels)
    return Hclg(S, 0, arc_off, arcs, final, tid2pdf, int(tid2pdf.max()) + 1)


def sample_utterance(g, n_words=6, seed=0, peak=6.0, noise=1.0, dur_p=0.5):
    """Random word sequence through `g` (from make_hclg) -> (loglikes [T,P], words,
    pdf alignment).  loglike = noise*N(0,1) + peak on the true pdf, minus logsumexp."""
    rng = np.random.default_rng(seed)
    h = g.start
    words, pdfs = [], []
    for _ in range(n_words):
        lo, hi = g.hist_pair_off[h], g.hist_pair_off[h + 1]
        if hi == lo:                     # history without words: back off
            h = 0
            lo, hi = g.hist_pair_off[0], g.hist_pair_off[1]
        k = int(rng.integers(lo, hi))
        w = int(g.pair_word[k])
        words.append(w + 1)
        ln = int(g.pair_len[k])
        for j in range(ln):
            u = int(g.chain_unit[int(g.pair_base[k, j])])
            pdfs.append(2 * u)           # forward pdf on entry
            d = int(rng.geometric(dur_p)) - 1
            pdfs.extend([2 * u + 1] * d)  # self-loop pdf
        h = int(g.word_next_hist[w])
    pdfs = np.asarray(pdfs, np.int64)
    T = pdfs.size
    ll = (noise * rng.standard_normal((T, g.num_pdfs))).astype(np.float32)
    ll[np.arange(T), pdfs] += peak
    m = ll.max(axis=1, keepdims=True)
    ll = ll - (m + np.log(np.exp(ll - m).sum(axis=1, keepdims=True)))
    return np.ascontiguousarray(ll, np.float32), words, pdfs


def sample_path(g, n_words=6, seed=0, dur_p=0.5):
    """The path of sample_utterance without its log-likelihood matrix: (words, pdf of every frame)."""
    rng = np.random.default_rng(seed)
    h = g.start
    words, pdfs = [], []
    for _ in range(n_words):
        lo, hi = g.hist_pair_off[h], g.hist_pair_off[h + 1]
        if hi == lo:
            h = 0
            lo, hi = g.hist_pair_off[0], g.hist_pair_off[1]
        k = int(rng.integers(lo, hi))
        w = int(g.pair_word[k])
        words.append(w + 1)
        ln = int(g.pair_len[k])
        units = g.chain_unit[g.pair_base[k, :ln]]
        durs = rng.geometric(dur_p, ln)                      # frames in each unit: entry + self-loops
        for u, d in zip(units.tolist(), durs.tolist()):
            pdfs.append(2 * u)
            pdfs.extend([2 * u + 1] * (d - 1))
        h = int(g.word_next_hist[w])
    return words, np.asarray(pdfs, np.int32)


def planted_loglikes_device(true_pdf, num_pdfs, peak, noise, seed=1):
    """[len(true_pdf) x num_pdfs] planted log-likelihoods generated on the device (kamd_synth_planted_loglikes_device);
    returns a kaldi_amd.decoder.DeviceMatrix-like object (ptr(row), rows, cols, download())."""
    import ctypes as C
    from . import decoder
    from ._lib import check, lib
    tp = np.ascontiguousarray(true_pdf, np.int32)
    d_tp = decoder.DeviceMatrix(tp.view(np.float32).reshape(-1, 1))           # (a byte container: int32 rows)
    m = decoder.DeviceMatrix.__new__(decoder.DeviceMatrix)
    m.rows, m.cols = int(tp.size), int(num_pdfs)
    m._d = lib().kamd_malloc(max(m.rows * m.cols * 4, 16))
    if not m._d:
        raise MemoryError("planted log-likelihoods: device allocation failed")
    check(lib().kamd_synth_planted_loglikes_device(m._d, m.rows, m.cols, m.cols, d_tp.ptr(0), float(peak), float(noise), int(seed), None))
    check(lib().kamd_device_synchronize())
    return m


def planted_loglikes_host(true_pdf, num_pdfs, peak, noise, seed=1):
    """The host twin of planted_loglikes_device (same statistics, another random stream): noise * N(0, 1) on every pdf and
    + peak on the pdf the planted path visits on that frame.  Seeded numpy: the same matrix on every machine."""
    tp = np.asarray(true_pdf, np.int64)
    rng = np.random.default_rng(seed)
    ll = rng.standard_normal((tp.size, int(num_pdfs)), dtype=np.float32)
    ll *= np.float32(noise)
    ll[np.arange(tp.size), tp] += np.float32(peak)
    return ll


def headline_sample(g, n_utts, total=2620, peak=8.3, noise=3.0):
    """A duration-stratified sample of the bench's planted test set (bench.planted_testset: utterance u = round(3 x seconds)
    words through the graph, seed 900000 + u), every (total // n_utts)-th utterance of the duration-sorted set, longest first:
    [(u, seconds, words, pdf path)].  Log-likelihoods: planted_loglikes_host(path, g.num_pdfs, peak, noise, seed=5000 + u)."""
    durs = utterance_durations(total, seed=1, mu=6.2)
    order = np.argsort(durs)
    k = max(1, total // n_utts)
    out = []
    for u in sorted((int(x) for x in order[k // 2::k]), key=lambda x: -durs[x]):
        words, path = sample_path(g, max(1, int(round(float(durs[u]) * 3.0))), seed=900000 + u)
        out.append((u, float(durs[u]), [int(w) for w in words], path))
    return out


def random_loglikes(T, P, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(scale * rng.standard_normal((T, P)), np.float32)


def make_wave(seconds, seed=0, samp_freq=16000):
    """Synthetic 16 kHz utterance: band-limited noise + a few drifting sinusoids,
    amplitude ~0.3*32768 (Kaldi keeps int16-scale floats, feat/wave-reader.cc:272)."""
    rng = np.random.default_rng(seed)
    n = int(seconds * samp_freq)
    t = np.arange(n) / samp_freq
    x = rng.standard_normal(n)
    k = np.array([0.25, 0.5, 0.25])
    x = np.convolve(x, k, mode="same")
    for _ in range(4):
        f0 = rng.uniform(100, 3500)
        x += 2.0 * np.sin(2 * np.pi * (f0 * t + 40 * np.sin(2 * np.pi * rng.uniform(1, 4) * t)))
    env = 0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2, 5) * t + rng.uniform(0, 6))
    x = x * env
    x = x / np.abs(x).max() * (0.3 * 32768)
    return np.round(x).astype(np.float32)


def make_waves_fast(durations, seed=0, samp_freq=16000):
    """A test set's worth of synthetic utterances in seconds instead of minutes: every utterance
    is a window of one long make_wave-style signal, with its own gain and one extra tone
    (the content only has to exercise the feature / nnet kernels; the search load of the bench
    is set by the calibration of the output layer, not by the audio)."""
    rng = np.random.default_rng(seed)
    durations = np.asarray(durations, np.float64)
    pool = make_wave(min(120.0, float(durations.max()) + 60.0), seed=seed + 1, samp_freq=samp_freq)
    out = []
    for d in durations:
        n = int(d * samp_freq)
        start = int(rng.integers(0, pool.size - n)) if pool.size > n else 0
        t = np.arange(n, dtype=np.float32) / np.float32(samp_freq)
        w = pool[start:start + n] * np.float32(rng.uniform(0.5, 1.0))
        w = w + np.float32(1500.0) * np.sin(np.float32(2 * np.pi * rng.uniform(150, 3000)) * t, dtype=np.float32)
        out.append(np.round(w).astype(np.float32))
    return out


def utterance_durations(n, seed=1, mu=7.0, sigma=0.6, lo=1.0, hi=35.0):
    """LibriSpeech-like durations: lognormal(ln 7 s, 0.6) clipped to [1, 35] s."""
    rng = np.random.default_rng(seed)
    return np.clip(rng.lognormal(np.log(mu), sigma, n), lo, hi)
