"""Token pre-selection of the work-queue lanes (kamd_decoder_set_token_preselection; decoder.hip InsertEmitted /
FindSkipped): on a frame that records several times max_active candidates the lane inserts only the candidates under a
bound that provably contains the next frame's max-active cutoff (+ those whose target has epsilon arcs) and turns the
left-out candidates into links where their target exists.  What must hold: the raw lattice, the best path, the final costs
and six of the seven work counters equal the oracle's bit for bit, exactly as without it; N_tok counts the tokens the lane
inserted (never more than the oracle created); and the same decoder with the switch off counts every token again."""
import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from oracle import orc
from tests.util import assert_work_counters, lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


def check_set(g, cfg, lls, mode, lanes=2, sizes=None):
    bd = decoder.BatchDecoder(decoder.Graph(g), cfg, sizes or abi.DecoderSizes(3, 1 << 16, 3 << 19, 3 << 20, 256))
    bd.SetSearchMode(mode)
    seen = {}
    for on in (True, False):
        bd.SetTokenPreselection(on)
        lats, recs, _ = bd.decode_queue(lls, resident_lanes=lanes)
        n_pre = 0
        for u, ll in enumerate(lls):
            o = orc.Decoder(g, cfg, mode)
            o.Decode(ll)
            lo = o.GetRawLattice()
            what = "preselection %s, utt %d, mode %d" % (on, u, mode)
            assert recs[u].status == 1 and recs[u].error == 0 and recs[u].n_frames == ll.shape[0], what
            if not on:
                assert recs[u].n_preselected == 0, what
            n_pre += recs[u].n_preselected
            if lo is None:
                assert lats[u] is None, what
                continue
            assert lattices_equal(lats[u], lo), what + ": " + lattice_diff(lats[u], lo)
            assert_work_counters(recs[u], o.counters(), err_msg=what)
            assert recs[u].final_relative_cost == o.FinalRelativeCost(), what
            bn, bo = decoder.lattice_best_path(lats[u]), lo.best_path()
            assert bn["words"].tolist() == bo["words"].tolist() and bn["graph_cost"] == bo["graph_cost"] and bn["acoustic_cost"] == bo["acoustic_cost"], what
        seen[on] = n_pre
    return seen


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("max_active,min_active", [(60, 0), (150, 20), (400, 0)])
def test_preselected_frames_equal_the_oracle(mode, max_active, min_active):
    g = synth.make_hclg(num_units=50, vocab=600, n_hist=60, fanout=(8, 40), seed=11, self_loop_prob=0.5, lm_scale=0.3)
    cfg = abi.decoder_config_recipe()
    cfg.max_active, cfg.min_active, cfg.beam, cfg.lattice_beam = max_active, min_active, 14.0, 6.0
    lls = [synth.random_loglikes(30 + 7 * i, g.num_pdfs, seed=50 + i, scale=0.6 + 0.2 * i) for i in range(4)]
    lls += [synth.sample_utterance(g, n_words=4 + i, seed=900 + i, peak=4.0, noise=1.5)[0] for i in range(3)]
    seen = check_set(g, cfg, lls, mode)
    assert seen[True] > 0, "no frame was pre-selected: the test does not test"


@pytest.mark.parametrize("mode", [1, 2])
def test_preselection_with_epsilon_closures(mode):
    """random graphs with many epsilon arcs: targets created by the closure, candidates with epsilon-flagged targets"""
    hit = 0
    for seed in range(6):
        g = synth.make_random_graph(num_states=900, num_labels=30, mean_arcs=6.0, eps_frac=0.25, seed=100 + seed, final_frac=0.2)
        cfg = abi.decoder_config_recipe()
        cfg.max_active, cfg.min_active, cfg.beam, cfg.lattice_beam = 50, 0, 16.0, 5.0
        lls = [synth.random_loglikes(25 + 5 * i, g.num_pdfs, seed=7 * seed + i, scale=0.5 + 0.3 * i) for i in range(3)]
        hit += check_set(g, cfg, lls, mode)[True]
    assert hit > 0


def test_preselection_is_off_for_a_live_decoder_and_on_a_calls_last_frame():
    """AdvanceKernel lanes keep every token (GetRawLattice of a live decoder may read the lists): all seven counters"""
    g = synth.make_hclg(num_units=50, vocab=600, n_hist=60, fanout=(8, 40), seed=11, self_loop_prob=0.5, lm_scale=0.3)
    cfg = abi.decoder_config_recipe()
    cfg.max_active, cfg.min_active, cfg.beam, cfg.lattice_beam = 60, 0, 14.0, 6.0
    ll = synth.random_loglikes(40, g.num_pdfs, seed=3, scale=0.8)
    d = decoder.LatticeFasterDecoder(decoder.Graph(g), cfg, abi.DecoderSizes(1, 1 << 16, 1 << 20, 1 << 21, 256))
    d.Decode(ll)
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    np.testing.assert_array_equal(np.asarray(d.counters()[:7]), o.counters()[:7])
    assert lattices_equal(d.GetRawLattice(), o.GetRawLattice())
