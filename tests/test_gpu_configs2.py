"""BASELINE.json configs[2] at full model / graph size: the LibriSpeech TDNN-F 1d topology (1536 / 160, 17 layers,
P = 6000) and the tglarge-scale synthetic HCLG (31 M states, 69 M arcs: beyond the Infinity Cache), through the
test-set decoder (work queue, host tail).  Sampled utterances must equal the oracle bit for bit given the device's
log-likelihoods (both search modes), the order-faithful mode 0 must give the same 1-best, nnet rows agree to 1e-4 of
the output scale, and what this size exercises must actually have been taken: a 24 KB log-likelihood row that is DMA'd
into LDS whole, one frame ahead (rows wider than the staging area: tests/test_gpu_decoder.py, P = 9000), and frames with
more tokens than the level-1 (LDS) table holds."""
import ctypes as C

import numpy as np
import pytest

from kaldi_amd import abi, batch, decoder, nnet, synth
from kaldi_amd._lib import lib
from oracle import orc
from tests.util import assert_work_counters, lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    g = synth.make_hclg(num_units=3000, vocab=200000, n_hist=160000, fanout=(12, 64), pron_len=(3, 7), seed=2,
                        self_loop_prob=0.5, lm_scale=0.3)
    model = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs)
    # the bench's calibration of the random output layer (search load), done once on the device
    import bench
    bench.calibrate(model, 1.9)
    durs = np.asarray([1.0, 1.2, 1.6, 2.0, 2.3, 2.9, 3.4, 4.0, 4.6, 5.2, 1.1, 1.4, 1.8, 2.6, 3.1, 3.7, 6.0, 0.9, 2.2, 5.6])
    waves = synth.make_waves_fast(durs, seed=77)
    G = decoder.Graph(g)
    return g, G, model, waves


@pytest.mark.parametrize("mode", [2, 1])
def test_configs2_queue_decode_equals_oracle(world, mode):
    g, G, model, waves = world
    assert g.num_states > 3e7 and g.num_arcs > 6e7 and g.num_pdfs == 6000
    cfg = abi.decoder_config_recipe()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, G, cfg, max_seconds=7.0, resident_lanes=8, host_threads=4,
                                determinize=True, keep_raw_lattices=True, hash_capacity=1 << 20, search_mode=mode,
                                nnet_pass_frames=1500)
    n_lds, tbl = C.c_int32(), C.c_int32()
    assert lib().kamd_decoder_lds_layout(bd.dec._dec, C.byref(n_lds), C.byref(tbl)) == 0
    assert n_lds.value == g.num_pdfs               # the whole row of P = 6000 is staged in LDS beside the tables
    bd.load(waves)
    st = bd.run()
    assert st.n_failed == 0 and st.nnet_passes > 1, [(u, bd.record(u).error, bd.record(u).n_frames) for u in range(len(waves)) if bd.output(u) is None]
    max_tok = 0
    for u in (0, 3, 7, 16, 19):
        ll = bd.loglikes(u)
        o = orc.Decoder(g, cfg, mode)
        o.Decode(ll)
        lo, lat = o.GetRawLattice(), bd.raw_lattice(u)
        assert lattices_equal(lat, lo), "utt %d: %s" % (u, lattice_diff(lat, lo))
        out, bo = bd.output(u), lo.best_path()
        assert out["words"].tolist() == bo["words"].tolist() and out["alignment"].tolist() == bo["alignment"].tolist()
        assert out["graph_cost"] == bo["graph_cost"] and out["acoustic_cost"] == bo["acoustic_cost"]
        assert_work_counters(out["record"], o.counters())
        max_tok = max(max_tok, int(o.trace()[0].max()))
        f = orc.Decoder(g, cfg, 0)                       # the reference's own order-dependent search
        f.Decode(ll)
        bf = f.GetRawLattice().best_path()
        assert bf["words"].tolist() == out["words"].tolist()
        assert abs((bf["graph_cost"] + bf["acoustic_cost"]) - (out["graph_cost"] + out["acoustic_cost"])) < 1e-3
        assert bd.compact_lattice(u).num_states > 0
    assert max_tok > tbl.value // 2                      # frames beyond the half-region table were decoded (whole-region frames)


def test_configs2_nnet_rows_match_the_cpu_port(world):
    g, G, model, waves = world
    cfg = abi.decoder_config_recipe()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, G, cfg, max_seconds=7.0, resident_lanes=4, host_threads=2,
                                determinize=False, hash_capacity=1 << 20)
    bd.load(waves[:6])
    bd.run()
    for u in (0, 5):
        feats = orc.mfcc(abi.mfcc_opts_hires(), waves[u])
        want = orc.nnet_forward(model, feats)
        got = bd.loglikes(u)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()) + 2e-3      # + the feature tolerance through the net


def _check_against_oracle(g, cfg, bd, ll_of, utts, mode, what):
    """Raw lattice, 1-best and the work counters of the sampled utterances against the oracle in the device's search mode, bit
    for bit; the order-faithful mode 0 (the reference restated) must give the same words."""
    seen = {"max_tok": 0, "level2": 0, "preselected": 0}
    for u in utts:
        ll = ll_of(u)
        o = orc.Decoder(g, cfg, mode)
        o.Decode(ll)
        lo, lat = o.GetRawLattice(), bd.raw_lattice(u)
        assert lattices_equal(lat, lo), "%s utt %d: %s" % (what, u, lattice_diff(lat, lo))
        out, bo = bd.output(u), lo.best_path()
        assert out["words"].tolist() == bo["words"].tolist() and out["alignment"].tolist() == bo["alignment"].tolist()
        assert out["graph_cost"] == bo["graph_cost"] and out["acoustic_cost"] == bo["acoustic_cost"]
        assert_work_counters(out["record"], o.counters())
        seen["max_tok"] = max(seen["max_tok"], int(o.trace()[0].max()))
        seen["level2"] += int(out["record"].counters[7])
        seen["preselected"] += int(out["record"].n_preselected)
        f = orc.Decoder(g, cfg, 0)                       # the reference's own order-dependent search
        f.Decode(ll)
        bf = f.GetRawLattice().best_path()
        seen.setdefault("mode0_same_words", []).append(bf["words"].tolist() == out["words"].tolist())
        seen.setdefault("mode0_cost_gap", []).append(abs((bf["graph_cost"] + bf["acoustic_cost"]) - (out["graph_cost"] + out["acoustic_cost"])))
    return seen


@pytest.mark.parametrize("ll_std", [1.4, 1.2])
def test_configs2_bench_loads_with_a_long_utterance(world, ll_std):
    """The loads bench.py reports (the token-matched spread 1.4 and the saturated 1.2, max-active binding), one utterance
    of 20 s among short ones, full graph: frames beyond the level-1 table (level-2 entries in HBM), frames beyond the
    finalize sweep's LDS working set, the work-queue lane's dropped tokens -- bit-exact against oracle mode 2."""
    g, G, model0, _ = world
    import bench
    model = nnet.tdnnf_librispeech(num_pdfs=g.num_pdfs)
    bench.calibrate(model, ll_std)
    waves = synth.make_waves_fast(np.asarray([20.0, 1.5, 2.5, 3.0]), seed=5)
    cfg = abi.decoder_config_recipe()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, G, cfg, max_seconds=21.0, resident_lanes=4, host_threads=4,
                                determinize=True, keep_raw_lattices=True, hash_capacity=1 << 20, search_mode=2, tokens_per_frame=11000)
    bd.load(waves)
    st = bd.run()
    assert st.n_failed == 0
    seen = _check_against_oracle(g, cfg, bd, bd.loglikes, (0, 2), 2, "spread %.1f" % ll_std)
    assert seen["max_tok"] > 16384 and seen["level2"] > 0          # beyond the whole level-1 region: level-2 entries were decoded
    assert seen["preselected"] > 0                                 # ... and frames whose inserts were pre-selected (InsertEmitted<KAMD_PS_FIRST>)
    assert all(seen["mode0_same_words"]), seen


def test_configs2_planted_transcripts_with_the_ivector_model(world):
    """bench.py's headline path: the model WITH the i-vector input (chunked, online i-vectors from the device extractor), the
    search reading planted log-likelihoods (kamd_batch_decoder_set_loglike_override), a 20 s utterance of ~60 words among
    short ones.  The device's lattices equal oracle mode 2 on the same planted rows bit for bit, counters included; the
    transcripts come back (a planted path is decodable by construction at this noise level)."""
    g, G, _, _ = world
    import bench
    from argparse import Namespace
    args = Namespace(workload="librispeech", graph="tglarge", output_scale=1.0, ll_std_ivectors=0.96, ll_std=1.4)
    model, ie = bench.ivector_variant(args, g)
    durs = np.asarray([20.0, 2.0, 3.0, 1.2])
    pset = bench.planted_testset(g, durs, list(range(durs.size)), synth)
    cfg = abi.decoder_config_recipe()
    bd = batch.NnetBatchDecoder(abi.mfcc_opts_hires(), model, G, cfg, max_seconds=pset["max_seconds"], resident_lanes=4, host_threads=4,
                                determinize=True, keep_raw_lattices=True, hash_capacity=1 << 20, search_mode=2, tokens_per_frame=11000,
                                nnet_pass_frames=4000)
    bd.set_ivector_extractor(ie, 50)
    bd.load_host(pset["waves"])
    assert np.array_equal(bd.output_frames(), pset["frames"])
    planted = synth.planted_loglikes_device(np.concatenate([p for _, p in pset["paths"]]), g.num_pdfs, 8.3, 3.0, seed=5)
    bd.set_loglike_override(planted.ptr(0))
    st = bd.run()
    assert st.n_failed == 0 and st.ivector_ms > 0
    ro = pset["row_off"]
    ll_of = lambda u: bench.planted_rows(planted, ro[u], ro[u + 1] - ro[u], g.num_pdfs)      # noqa: E731
    seen = _check_against_oracle(g, cfg, bd, ll_of, (0, 1), 2, "planted")
    assert seen["level2"] > 0 and seen["preselected"] > 0
    words = bd.output(0)["words"].tolist()
    ref = list(pset["paths"][0][0])
    assert len(words) > 30 and bench._edit_distance(ref, words) <= 0.25 * len(ref)
    print("planted: mode-0 same words %s, cost gaps %s, max tokens per frame %d" % (seen["mode0_same_words"], seen["mode0_cost_gap"], seen["max_tok"]))
