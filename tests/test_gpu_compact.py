"""kamd_decoder_compact = PruneActiveTokens in the middle of an utterance (decoder/lattice-faster-decoder.cc:519-546):
the lane's arenas lose what the final backward sweep would drop anyway, decoding goes on, and the final lattice is
bit for bit the one an uncompacted decode gives -- including when the uncompacted decode would not have fitted."""
import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from kaldi_amd._lib import KamdError
from oracle import orc
from tests.util import lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


def _workload(seed=5, n_words=14, peak=3.0):
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=3)
    ll, _, _ = synth.sample_utterance(g, n_words=n_words, seed=seed, peak=peak, noise=1.5)
    return g, ll


@pytest.mark.parametrize("mode", [1, 2])
def test_compacting_after_every_chunk_changes_nothing_but_the_memory(mode):
    g, ll = _workload()
    cfg = abi.decoder_config_recipe()
    G = decoder.Graph(g)
    sz = abi.DecoderSizes(1, 1 << 14, 1 << 20, 1 << 21, 1024)
    ref = decoder.LatticeFasterDecoder(G, cfg, sz)
    ref.SetSearchMode(mode)
    ref.Decode(ll)
    want = ref.GetRawLattice()
    ref.InitDecoding()
    ref.AdvanceDecoding(ll)
    used_plain = ref.usage()
    d = decoder.LatticeFasterDecoder(G, cfg, sz)
    d.SetSearchMode(mode)
    d.InitDecoding()
    step, shrunk = 11, 0
    for t in range(0, ll.shape[0], step):
        d.AdvanceDecoding(ll[t:t + step])
        before = d.usage()
        bp_before = decoder.partial_best_path(d._dec, d.lane, False)
        d.PruneActiveTokens()
        after = d.usage()
        assert after[0] <= before[0] and after[2] <= before[2]
        shrunk += int(after[0] < before[0])
        bp_after = decoder.partial_best_path(d._dec, d.lane, False)
        assert bp_before["alignment"].tolist() == bp_after["alignment"].tolist() and bp_before["words"].tolist() == bp_after["words"].tolist()
        assert bp_before["graph_cost"] == bp_after["graph_cost"] and bp_before["acoustic_cost"] == bp_after["acoustic_cost"]
        # compacting twice in a row: the second one finds nothing to drop
        d.PruneActiveTokens()
        assert d.usage() == after
    assert shrunk > 3 and d.usage()[0] < used_plain[0] // 2
    d.FinalizeDecoding()
    got = d.GetRawLattice()
    assert lattices_equal(got, want), lattice_diff(got, want)
    o = orc.Decoder(g, cfg, mode)
    o.Decode(ll)
    assert lattices_equal(got, o.GetRawLattice())


def test_an_utterance_that_does_not_fit_uncompacted_decodes_with_compaction():
    g, ll = _workload(seed=9, n_words=30, peak=2.5)
    cfg = abi.decoder_config_recipe()
    G = decoder.Graph(g)
    big = decoder.LatticeFasterDecoder(G, cfg, abi.DecoderSizes(1, 1 << 14, 1 << 21, 1 << 22, 2048))
    big.Decode(ll)
    want = big.GetRawLattice()
    big.InitDecoding(); big.AdvanceDecoding(ll)
    tok_all, _, lnk_all, _ = big.usage()
    small = abi.DecoderSizes(1, 1 << 14, max(4096, tok_all // 3), max(8192, lnk_all // 3), 2048)
    d = decoder.LatticeFasterDecoder(G, cfg, small)
    with pytest.raises(KamdError):                 # the arenas hold a third of what the utterance creates
        d.Decode(ll)
    d = decoder.LatticeFasterDecoder(G, cfg, small)
    d.InitDecoding()
    for t in range(0, ll.shape[0], 10):
        d.AdvanceDecoding(ll[t:t + 10])
        tu, tc, lu, lc = d.usage()
        if tu > tc // 3 or lu > lc // 3:
            d.PruneActiveTokens()
    d.FinalizeDecoding()
    assert lattices_equal(d.GetRawLattice(), want)


@pytest.mark.parametrize("step", [5, 13])
def test_compaction_through_both_finalize_modes(step):
    """frames of tens of thousands of tokens (the sweep's HBM mode) alternating with frames of a few (LDS mode), both
    transitions, compaction landing on either kind"""
    from tests.test_gpu_decoder import _mixed_load, sizes
    g = synth.make_hclg(num_units=200, vocab=3000, n_hist=300, fanout=(10, 60), seed=0)
    ll = _mixed_load(g)
    cfg = abi.decoder_config_recipe()
    cfg.max_active = 30000
    G = decoder.Graph(g)
    sz = sizes(hash_cap=1 << 16, toks=1 << 21, links=1 << 23)
    ref = decoder.LatticeFasterDecoder(G, cfg, sz)
    ref.Decode(ll)
    want = ref.GetRawLattice()
    d = decoder.LatticeFasterDecoder(G, cfg, sz)
    d.InitDecoding()
    peak = 0
    for t in range(0, ll.shape[0], step):
        d.AdvanceDecoding(ll[t:t + step])
        peak = max(peak, d.usage()[0])
        d.PruneActiveTokens()
    assert peak > 20000
    d.FinalizeDecoding()
    got = d.GetRawLattice()
    assert lattices_equal(got, want), lattice_diff(got, want)


def test_streams_compact_their_arenas_when_they_fill_up():
    """kamd_stream_batch: arenas a quarter of what the utterances create; with compaction at 50 % (the default) every
    stream decodes and its lattice is the offline one; with compaction off the streams overflow."""
    from kaldi_amd import feat, nnet, online
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=1.2)          # flat scores: many tokens survive
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    waves = [synth.make_wave(d_, seed=70 + i) for i, d_ in enumerate((4.0, 2.5, 3.1))]
    offline = []
    for w in waves:
        off = decoder.LatticeFasterDecoder(G, cfg, abi.DecoderSizes(1, 1 << 14, 1 << 21, 1 << 22, 1024))
        ll = N.Forward(feat.Mfcc(op).ComputeFeatures(w))
        off.Decode(ll)
        offline.append(off.GetRawLattice())
        off.InitDecoding(); off.AdvanceDecoding(ll)
        offline[-1].created = off.usage()
    S = len(waves)
    tok_cap = max(l.created[0] for l in offline) // 4
    lnk_cap = max(l.created[2] for l in offline) // 4
    sizes = abi.DecoderSizes(S, 1 << 14, S * tok_cap, S * lnk_cap, 1024)

    def run(fraction):
        sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=5.0, sizes=sizes)
        if fraction is not None:
            sb.set_compaction(fraction)
        sb.start(list(range(S)))
        pos = [0] * S
        step = int(0.18 * 16000)
        while any(pos[s] < waves[s].size for s in range(S)):
            live = [s for s in range(S) if pos[s] < waves[s].size]
            for s in live:
                sb.accept(s, waves[s][pos[s]:pos[s] + step], input_finished=pos[s] + step >= waves[s].size)
                pos[s] += step
            sb.advance(live)
        sb.finalize(list(range(S)))
        return sb
    sb = run(None)
    assert sb.num_compactions() >= S
    for s in range(S):
        assert lattices_equal(sb.raw_lattice(s), offline[s]), lattice_diff(sb.raw_lattice(s), offline[s])
    with pytest.raises(KamdError):
        run(0.0)


def test_compaction_at_the_edges():
    """right after InitDecoding (nothing decoded), after a single frame, twice in a row, and on a finalized lane (a no-op)"""
    g, ll = _workload(seed=3, n_words=5)
    cfg = abi.decoder_config_recipe()
    G = decoder.Graph(g)
    sz = abi.DecoderSizes(1, 1 << 14, 1 << 19, 1 << 20, 512)
    ref = decoder.LatticeFasterDecoder(G, cfg, sz)
    ref.Decode(ll)
    want = ref.GetRawLattice()
    d = decoder.LatticeFasterDecoder(G, cfg, sz)
    d.InitDecoding()
    d.PruneActiveTokens()
    assert d.NumFramesDecoded() == 0
    d.AdvanceDecoding(ll[:1])
    d.PruneActiveTokens(); d.PruneActiveTokens()
    d.AdvanceDecoding(ll[1:])
    d.FinalizeDecoding()
    got = d.GetRawLattice()
    assert lattices_equal(got, want), lattice_diff(got, want)
    d.PruneActiveTokens()                                  # finalized: nothing happens
    assert lattices_equal(d.GetRawLattice(), want)
    # an empty utterance: InitDecoding, compaction, FinalizeDecoding
    e = decoder.LatticeFasterDecoder(G, cfg, sz)
    e.InitDecoding(); e.PruneActiveTokens(); e.FinalizeDecoding()
    r = decoder.LatticeFasterDecoder(G, cfg, sz)
    r.InitDecoding(); r.FinalizeDecoding()
    a, b = e.GetRawLattice(), r.GetRawLattice()
    assert (a is None and b is None) or lattices_equal(a, b)
