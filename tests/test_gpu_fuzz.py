"""Randomised parity sweep: many small (graph, scores, config) triples, HIP decoder vs the
canonical oracle, bit-exact (the same comparison as tests/test_gpu_decoder.py)."""
import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from oracle import orc
from tests.test_gpu_decoder import assert_same, sizes
from tests.util import assert_work_counters

pytestmark = pytest.mark.gpu


def random_case(rng):
    kind = rng.integers(3)
    if kind == 0:
        g = synth.make_random_graph(num_states=int(rng.integers(20, 400)), num_labels=int(rng.integers(5, 60)),
                                    mean_arcs=float(rng.uniform(1.5, 5.0)), eps_frac=float(rng.uniform(0.0, 0.3)),
                                    seed=int(rng.integers(1 << 30)))
        ll = synth.random_loglikes(int(rng.integers(1, 60)), g.num_pdfs, seed=int(rng.integers(1 << 30)),
                                   scale=float(rng.uniform(0.3, 3.0)))
    else:
        g = synth.make_hclg(num_units=int(rng.integers(8, 60)), vocab=int(rng.integers(20, 400)),
                            n_hist=int(rng.integers(3, 50)), seed=int(rng.integers(1 << 30)),
                            self_loop_prob=float(rng.uniform(0.2, 0.7)), lm_scale=float(rng.uniform(0.1, 1.0)))
        if kind == 1:
            ll, _, _ = synth.sample_utterance(g, n_words=int(rng.integers(1, 8)), seed=int(rng.integers(1 << 30)),
                                              peak=float(rng.uniform(1.0, 8.0)), noise=float(rng.uniform(0.3, 2.0)))
        else:
            ll = synth.random_loglikes(int(rng.integers(1, 50)), g.num_pdfs, seed=int(rng.integers(1 << 30)),
                                       scale=float(rng.uniform(0.3, 2.0)))
    cfg = abi.decoder_config_recipe()
    cfg.beam = float(rng.choice([4.0, 8.0, 12.0, 15.0, 20.0]))
    cfg.lattice_beam = float(rng.choice([0.5, 2.0, 6.0, 8.0, 10.0]))
    cfg.max_active = int(rng.choice([40, 200, 1000, 7000, abi.INT32_MAX]))
    cfg.min_active = int(rng.choice([0, 5, 20, 200]))
    if cfg.min_active >= cfg.max_active:
        cfg.min_active = 0
    return g, ll, cfg


import os


@pytest.mark.parametrize("block", range(int(os.environ.get("KAMD_FUZZ_BLOCKS", "6"))))
def test_random_cases(block):
    rng = np.random.default_rng(int(os.environ.get("KAMD_FUZZ_SEED", "1234")) + block)   # soak runs: other seeds, more blocks
    for i in range(20):
        g, ll, cfg = random_case(rng)
        G = decoder.Graph(g)
        d = decoder.LatticeFasterDecoder(G, cfg, sizes(hash_cap=1 << 15, toks=1 << 20, links=1 << 21, frames=256))
        d.Decode(ll)
        o = orc.Decoder(g, cfg, 1)
        o.Decode(ll)
        try:
            assert_same(d, o)
        except AssertionError as e:
            raise AssertionError("block %d case %d (%d states, %d frames, beam %g lattice_beam %g max %d min %d): %s" % (
                block, i, g.num_states, ll.shape[0], cfg.beam, cfg.lattice_beam, cfg.max_active, cfg.min_active, e))


@pytest.mark.parametrize("block", range(int(os.environ.get("KAMD_FUZZ_BLOCKS", "6"))))
def test_random_cases_chunked_compacted_both_search_modes(block):
    """the round-2 paths under the same sweep: search mode 1 or 2, AdvanceDecoding in random chunks, PruneActiveTokens
    (kamd_decoder_compact) after random chunks -- still bit-exact against the oracle's one-shot decode in that mode"""
    rng = np.random.default_rng(int(os.environ.get("KAMD_FUZZ_SEED", "1234")) + 1000 + block)
    for i in range(20):
        g, ll, cfg = random_case(rng)
        mode = int(rng.integers(1, 3))
        G = decoder.Graph(g)
        d = decoder.LatticeFasterDecoder(G, cfg, sizes(hash_cap=1 << 15, toks=1 << 20, links=1 << 21, frames=256))
        d.SetSearchMode(mode)
        d.InitDecoding()
        t, n_compact = 0, 0
        while t < ll.shape[0]:
            n = int(rng.integers(1, 12))
            d.AdvanceDecoding(ll[t:t + n])
            t += n
            if rng.random() < 0.5:
                d.PruneActiveTokens(); n_compact += 1
        d.FinalizeDecoding()
        o = orc.Decoder(g, cfg, mode)
        o.Decode(ll)
        try:
            assert_same(d, o)
        except AssertionError as e:
            raise AssertionError("block %d case %d mode %d (%d states, %d frames, %d compactions, beam %g lattice_beam %g max %d min %d): %s" % (
                block, i, mode, g.num_states, ll.shape[0], n_compact, cfg.beam, cfg.lattice_beam, cfg.max_active, cfg.min_active, e))


@pytest.mark.parametrize("block", range(int(os.environ.get("KAMD_FUZZ_BLOCKS", "6"))))
def test_random_cases_through_the_work_queue(block):
    """the fused queue kernel (InitDecoding + AdvanceDecoding + FinalizeDecoding + lattice hand-off per task) under the
    same sweep: per random (graph, config) a few utterances through 1-3 resident lanes, each bit-exact against the oracle"""
    rng = np.random.default_rng(int(os.environ.get("KAMD_FUZZ_SEED", "1234")) + 2000 + block)
    for i in range(10):
        g, ll0, cfg = random_case(rng)
        mode = int(rng.integers(1, 3))
        lls = [ll0] + [synth.random_loglikes(int(rng.integers(1, 40)), g.num_pdfs, seed=int(rng.integers(1 << 30)),
                                             scale=float(rng.uniform(0.3, 2.5))) for _ in range(int(rng.integers(1, 4)))]
        bd = decoder.BatchDecoder(decoder.Graph(g), cfg, abi.DecoderSizes(3, 1 << 15, 3 << 19, 3 << 20, 256))
        bd.SetSearchMode(mode)
        lats, recs, _ = bd.decode_queue(lls, resident_lanes=int(rng.integers(1, 4)))
        for u, ll in enumerate(lls):
            o = orc.Decoder(g, cfg, mode)
            o.Decode(ll)
            lo = o.GetRawLattice()
            what = "block %d case %d utt %d mode %d (%d states, %d frames)" % (block, i, u, mode, g.num_states, ll.shape[0])
            assert recs[u].status == 1 and recs[u].error == 0 and recs[u].n_frames == ll.shape[0], what
            if lo is None:
                assert lats[u] is None, what
                continue
            from tests.util import lattice_diff, lattices_equal
            assert lattices_equal(lats[u], lo), what + ": " + lattice_diff(lats[u], lo)
            assert_work_counters(recs[u], o.counters(), err_msg=what)
            assert recs[u].final_relative_cost == o.FinalRelativeCost(), what


@pytest.mark.parametrize("block", range(int(os.environ.get("KAMD_FUZZ_STREAM_BLOCKS", "2"))))
def test_random_streams_with_compaction(block):
    """kamd_stream_batch_* under random chunkings and compaction thresholds: every stream's lattice equals the offline
    decode of the same waveform, however often its arena was compacted on the way"""
    from kaldi_amd import feat, nnet, online
    rng = np.random.default_rng(int(os.environ.get("KAMD_FUZZ_SEED", "1234")) + 3000 + block)
    g = synth.make_hclg(num_units=int(rng.integers(10, 30)), vocab=int(rng.integers(20, 80)), n_hist=int(rng.integers(4, 16)),
                        seed=int(rng.integers(1 << 30)))
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, output_scale=float(rng.uniform(1.0, 4.0)), seed=int(rng.integers(1 << 20)))
    N, G = decoder.Nnet(m), decoder.Graph(g)
    op, cfg = abi.mfcc_opts_hires(), abi.decoder_config_recipe()
    cfg.lattice_beam = float(rng.choice([2.0, 6.0, 8.0]))
    S = int(rng.integers(1, 5))
    waves = [synth.make_wave(float(rng.uniform(0.4, 3.0)), seed=int(rng.integers(1 << 30))) for _ in range(S)]
    want = []
    for w in waves:
        off = decoder.LatticeFasterDecoder(G, cfg, abi.DecoderSizes(1, 1 << 14, 1 << 21, 1 << 22, 1024))
        off.Decode(N.Forward(feat.Mfcc(op).ComputeFeatures(w)))
        want.append(off.GetRawLattice())
    sb = online.StreamBatch(op, N, G, cfg, S, max_seconds=4.0, sizes=abi.DecoderSizes(S, 1 << 14, S << 19, S << 20, 1024))
    sb.set_compaction(float(rng.choice([0.0, 0.002, 0.01, 0.05, 0.5])))
    sb.start(list(range(S)))
    pos = [0] * S
    while any(pos[s] < waves[s].size for s in range(S)):
        live = [s for s in range(S) if pos[s] < waves[s].size and rng.random() < 0.8]
        for s in live:
            n = int(rng.integers(400, 8000))
            sb.accept(s, waves[s][pos[s]:pos[s] + n], input_finished=pos[s] + n >= waves[s].size)
            pos[s] += n
        if live:
            sb.advance(live)
    sb.finalize(list(range(S)))
    from tests.util import lattice_diff, lattices_equal
    for s in range(S):
        got = sb.raw_lattice(s)
        assert lattices_equal(got, want[s]), "block %d stream %d (%d compactions): %s" % (block, s, sb.num_compactions(), lattice_diff(got, want[s]))
