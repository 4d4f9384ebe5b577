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
from tests.util import lattice_diff, lattices_equal

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
    assert st.n_failed == 0 and st.nnet_passes > 1
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
        np.testing.assert_array_equal(np.asarray(out["record"].counters[:7]), o.counters()[:7])
        max_tok = max(max_tok, int(o.trace()[0].max()))
        f = orc.Decoder(g, cfg, 0)                       # the reference's own order-dependent search
        f.Decode(ll)
        bf = f.GetRawLattice().best_path()
        assert bf["words"].tolist() == out["words"].tolist()
        assert abs((bf["graph_cost"] + bf["acoustic_cost"]) - (out["graph_cost"] + out["acoustic_cost"])) < 1e-3
        assert bd.compact_lattice(u).num_states > 0
    assert max_tok > tbl.value                           # frames that overflow the level-1 table into HBM were decoded


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
