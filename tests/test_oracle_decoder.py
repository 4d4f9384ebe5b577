"""Pins the decoder oracle (oracle/orc_decoder.cc) without the reference binary:
 (a) exact float64 Viterbi / forward-backward over the full trellis (no beam) must agree
     with the oracle's best path and with the token set that survives lattice pruning;
 (b) the order-faithful mode and the canonical (order-independent) mode must agree on
     1-best and on the final lattice wherever the reference is order independent;
 (c) a peaked synthetic decode must recover the truth transcript.
The reference has no decoder unit tests (src/decoder/Makefile: TESTFILES empty)."""
import numpy as np
import pytest

from kaldi_amd import abi, synth
from oracle import orc
from tests.util import lattice_diff, lattices_equal


def eps_closure(cost, g, eps_arcs):
    changed = True
    while changed:
        changed = False
        for (s, d, w) in eps_arcs:
            if cost[s] + w < cost[d]:
                cost[d] = cost[s] + w
                changed = True
    return cost


def trellis(g, ll):
    """float64 forward/backward over (frame, state); returns alpha, beta, best."""
    S, T = g.num_states, ll.shape[0]
    src = np.repeat(np.arange(S), np.diff(g.arc_off))
    a = g.arcs
    emit = [(int(s), int(d), float(w), int(g.tid2pdf[i])) for s, d, w, i in
            zip(src, a["nextstate"], a["weight"], a["ilabel"]) if i != 0]
    eps = [(int(s), int(d), float(w)) for s, d, w, i in
           zip(src, a["nextstate"], a["weight"], a["ilabel"]) if i == 0]
    alpha = np.full((T + 1, S), np.inf)
    alpha[0, g.start] = 0.0
    eps_closure(alpha[0], g, eps)
    for f in range(T):
        for (s, d, w, p) in emit:
            c = alpha[f, s] + w - float(ll[f, p])
            if c < alpha[f + 1, d]:
                alpha[f + 1, d] = c
        eps_closure(alpha[f + 1], g, eps)
    beta = np.full((T + 1, S), np.inf)
    beta[T] = g.final.astype(np.float64)
    any_final = np.isfinite(alpha[T] + beta[T]).any()
    if not any_final:
        beta[T] = 0.0
    reps = [(d, s, w) for (s, d, w) in eps]

    def back_eps(b):
        changed = True
        while changed:
            changed = False
            for (s, d, w) in eps:
                if b[d] + w < b[s]:
                    b[s] = b[d] + w
                    changed = True
    back_eps(beta[T])
    for f in range(T - 1, -1, -1):
        for (s, d, w, p) in emit:
            c = beta[f + 1, d] + w - float(ll[f, p])
            if c < beta[f, s]:
                beta[f, s] = c
        back_eps(beta[f])
    best = (alpha[T] + beta[T]).min()
    return alpha, beta, best


def wide_cfg(lattice_beam=4.0):
    c = abi.decoder_config_default()
    c.beam, c.max_active, c.min_active, c.lattice_beam = 1000.0, abi.INT32_MAX, 0, lattice_beam
    return c


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("mode", [0, 1])
def test_best_path_and_lattice_vs_exact_trellis(seed, mode):
    g = synth.make_random_graph(num_states=60, num_labels=12, mean_arcs=2.5, seed=seed,
                                final_frac=0.3)
    T = 7
    ll = synth.random_loglikes(T, g.num_pdfs, seed=100 + seed, scale=1.5)
    alpha, beta, best = trellis(g, ll)
    if not np.isfinite(best):
        pytest.skip("no complete path")
    cfg = wide_cfg(4.0)
    d = orc.Decoder(g, cfg, mode)
    d.Decode(ll)
    lat = d.GetRawLattice()
    bp = lat.best_path()
    assert bp is not None
    assert abs(bp["graph_cost"] + bp["acoustic_cost"] - best) < 1e-3
    # surviving tokens == {alpha+beta-best <= lattice_beam}, away from the knife edge
    extra = alpha + beta - best
    got = set(zip(lat.frame.tolist(), lat.hclg.tolist()))
    for f in range(T + 1):
        for s in range(g.num_states):
            e = extra[f, s]
            if e <= cfg.lattice_beam - 1e-3:
                assert (f, s) in got, (f, s, e)
            elif e > cfg.lattice_beam + 1e-3:
                assert (f, s) not in got, (f, s, e)
    # every lattice arc lies on a path within lattice_beam
    for a in lat.arcs:
        f0, s0 = int(lat.frame[a["src"]]), int(lat.hclg[a["src"]])
        f1, s1 = int(lat.frame[a["dst"]]), int(lat.hclg[a["dst"]])
        assert f1 - f0 == (1 if a["ilabel"] != 0 else 0)
        through = alpha[f0, s0] + float(a["graph_cost"]) + float(a["acoustic_cost"]) + beta[f1, s1]
        assert through - best <= cfg.lattice_beam + 1e-3


def test_forward_costs_match_trellis_with_offsets():
    g = synth.make_random_graph(num_states=40, num_labels=10, seed=11, final_frac=0.4)
    ll = synth.random_loglikes(6, g.num_pdfs, seed=12)
    alpha, beta, best = trellis(g, ll)
    d = orc.Decoder(g, wide_cfg(1000.0), 1)
    d.Decode(ll)
    lat = d.GetRawLattice()
    _, _, off = d.trace()
    cum = np.concatenate([[0.0], np.cumsum(off.astype(np.float64))])
    # tot_cost carries the running cost_offsets (lattice-faster-decoder.cc:760,793)
    for i in range(lat.frame.size):
        f, s = int(lat.frame[i]), int(lat.hclg[i])
        assert abs(float(lat.cost[i]) - (alpha[f, s] + cum[f])) < 1e-3


@pytest.mark.parametrize("seed", range(4))
def test_faithful_equals_canonical_unsaturated(seed):
    """With beam-only pruning (no max-active) and min_active not binding, the reference's
    order-dependent extras never get outgoing links, so the final lattices coincide."""
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=seed)
    ll, words, _ = synth.sample_utterance(g, n_words=5, seed=seed, peak=7.0)
    cfg = abi.decoder_config_recipe()
    cfg.min_active = 0
    lats = []
    for mode in (0, 1):
        d = orc.Decoder(g, cfg, mode)
        d.Decode(ll)
        lats.append(d.GetRawLattice())
    assert lattices_equal(lats[0], lats[1]), lattice_diff(lats[0], lats[1])
    bp = lats[1].best_path()
    assert bp["words"].tolist() == words


@pytest.mark.parametrize("seed", range(3))
def test_faithful_vs_canonical_saturated_same_best_path(seed):
    """max-active binding every frame: token sets may differ by order-dependent extras,
    the 1-best (words and cost) must not."""
    g = synth.make_hclg(num_units=16, vocab=80, n_hist=10, seed=20 + seed)
    ll = synth.random_loglikes(25, g.num_pdfs, seed=seed, scale=0.7)
    cfg = abi.decoder_config_recipe()
    cfg.max_active, cfg.min_active = 150, 20
    res = []
    for mode in (0, 1):
        d = orc.Decoder(g, cfg, mode)
        d.Decode(ll)
        lat = d.GetRawLattice()
        res.append((lat, lat.best_path(), d.trace()))
    assert (res[0][2][0] > cfg.max_active).any()      # the constraint really binds
    b0, b1 = res[0][1], res[1][1]
    assert b0["words"].tolist() == b1["words"].tolist()
    assert abs((b0["graph_cost"] + b0["acoustic_cost"]) - (b1["graph_cost"] + b1["acoustic_cost"])) < 1e-3
    # cutoffs are order independent: identical per-frame traces of cost offsets
    np.testing.assert_array_equal(res[0][2][2], res[1][2][2])


def test_intermediate_pruning_is_conservative():
    """PruneActiveTokens every 25 frames (reference) vs pruning only at the end: same final
    lattice, also when max-active binds (4 prune passes over 110 frames)."""
    g = synth.make_hclg(num_units=60, vocab=400, n_hist=80, seed=3, self_loop_prob=0.5, lm_scale=0.1)
    ll, _, _ = synth.sample_utterance(g, n_words=14, seed=2, peak=3.0, noise=1.3)
    assert ll.shape[0] > 75
    lats = []
    for interval in (25, 1 << 30):
        cfg = abi.decoder_config_recipe()
        cfg.max_active, cfg.min_active, cfg.prune_interval = 1500, 100, interval
        d = orc.Decoder(g, cfg, 0)
        d.Decode(ll)
        lats.append(d.GetRawLattice())
        assert (d.trace()[0] > cfg.max_active).any()
    assert lattices_equal(lats[0], lats[1]), lattice_diff(lats[0], lats[1])


def test_truth_recovered_and_lattice_contains_it():
    g = synth.make_hclg(num_units=40, vocab=120, n_hist=20, seed=5)
    ll, words, pdfs = synth.sample_utterance(g, n_words=8, seed=9, peak=8.0)
    d = orc.Decoder(g, abi.decoder_config_recipe(), 1)
    d.Decode(ll)
    bp = d.GetRawLattice().best_path()
    assert bp["words"].tolist() == words
    assert g.tid2pdf[bp["alignment"]].tolist() == pdfs.tolist()
    assert np.isfinite(d.FinalRelativeCost())   # a final state was reached


def test_advance_in_pieces_equals_one_shot():
    """AdvanceDecoding(max_num_frames) chunking (lattice-faster-decoder.cc:615-631)."""
    g = synth.make_hclg(num_units=24, vocab=60, n_hist=12, seed=3)
    ll, _, _ = synth.sample_utterance(g, n_words=6, seed=4)
    cfg = abi.decoder_config_recipe()
    for mode in (0, 1):
        a = orc.Decoder(g, cfg, mode)
        a.Decode(ll)
        b = orc.Decoder(g, cfg, mode)
        b.InitDecoding()
        for i in range(0, ll.shape[0], 7):
            b.AdvanceDecoding(ll[i:i + 7])
        b.FinalizeDecoding()
        assert lattices_equal(a.GetRawLattice(), b.GetRawLattice())


def test_min_active_and_max_active_cutoffs():
    """GetCutoff branches (lattice-faster-decoder.cc:693-722) on a hand-checkable case."""
    g = synth.make_random_graph(num_states=200, num_labels=20, mean_arcs=4, seed=7, final_frac=0.5)
    ll = synth.random_loglikes(12, g.num_pdfs, seed=8, scale=3.0)
    cfg = abi.decoder_config_default()
    cfg.beam, cfg.max_active, cfg.min_active, cfg.lattice_beam = 2.0, 30, 10, 3.0
    d = orc.Decoder(g, cfg, 1)
    d.Decode(ll)
    ntok, cutoff, off = d.trace()
    assert (ntok > 0).all()
    # ntok <= min_active leaves min_active_cutoff at +inf, which is > beam_cutoff, so the
    # reference returns +inf with an infinite adaptive beam (lattice-faster-decoder.cc:704-718)
    np.testing.assert_array_equal(np.isinf(cutoff), ntok <= cfg.min_active)
    assert d.GetRawLattice() is not None


def test_empty_and_single_frame():
    g = synth.make_hclg(num_units=8, vocab=10, n_hist=3, seed=1)
    d = orc.Decoder(g, abi.decoder_config_recipe(), 1)
    d.InitDecoding()
    assert d.NumFramesDecoded() == 0
    lat = d.GetRawLattice()            # num_frames == 0: the reference asserts num_frames > 0
    ll = synth.random_loglikes(1, g.num_pdfs, seed=2)
    d.AdvanceDecoding(ll)
    d.FinalizeDecoding()
    assert d.NumFramesDecoded() == 1
