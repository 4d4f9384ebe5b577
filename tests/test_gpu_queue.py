"""Work-queue decoding (kamd_decoder_queue_*): persistent lanes that pull utterances until the
queue is empty must give, for every utterance, exactly what a decoder of its own gives -- the
oracle's canonical lattice bit for bit, the same counters, the same best path -- whatever the
number of resident lanes, the hand-out order, and whichever lane happened to decode it."""
import ctypes as C

import numpy as np
import pytest

from kaldi_amd import abi, decoder, synth
from kaldi_amd._lib import KamdError, lib
from oracle import orc
from tests.util import assert_work_counters, lattice_diff, lattices_equal

pytestmark = pytest.mark.gpu


def oracle_lattice(g, cfg, ll):
    o = orc.Decoder(g, cfg, 1)
    o.Decode(ll)
    return o


def make_set(g, n, seed0=0, peaked=True):
    lls = []
    for i in range(n):
        if peaked:
            ll, _, _ = synth.sample_utterance(g, n_words=1 + (7 * i) % 9, seed=seed0 + i, peak=5.0, noise=1.5)
        else:
            ll = synth.random_loglikes(5 + (13 * i) % 60, g.num_pdfs, seed=seed0 + i, scale=1.2)
        lls.append(ll)
    return lls


@pytest.mark.parametrize("lanes", [1, 3, 8])
def test_queue_equals_oracle(lanes):
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=5)
    cfg = abi.decoder_config_recipe()
    lls = make_set(g, 21, seed0=100)
    G = decoder.Graph(g)
    bd = decoder.BatchDecoder(G, cfg, abi.DecoderSizes(8, 1 << 14, 1 << 18, 1 << 19, 512))
    lats, recs, ms = bd.decode_queue(lls, resident_lanes=lanes)
    used = set()
    for i, ll in enumerate(lls):
        o = oracle_lattice(g, cfg, ll)
        lo = o.GetRawLattice()
        assert recs[i].status == 1 and recs[i].error == 0
        assert recs[i].n_frames == ll.shape[0]
        assert lattices_equal(lats[i], lo), "utt %d: %s" % (i, lattice_diff(lats[i], lo))
        assert_work_counters(recs[i], o.counters())
        assert recs[i].final_relative_cost == o.FinalRelativeCost()
        bn, bo = decoder.lattice_best_path(lats[i]), lo.best_path()
        assert bn["words"].tolist() == bo["words"].tolist()
        assert bn["alignment"].tolist() == bo["alignment"].tolist()
        assert bn["graph_cost"] == bo["graph_cost"] and bn["acoustic_cost"] == bo["acoustic_cost"]
        used.add(recs[i].lane)
    assert len(used) <= lanes and max(used) < lanes
    if lanes > 1:
        assert len(used) > 1            # the work really was spread
    assert sorted(bd.queue_order.tolist()) == list(range(len(lls)))


def test_queue_order_independent_and_reusable():
    """Same results for a different hand-out order and on a second launch of the same decoder
    (lanes carry epsilon-closure stamps and table state from one utterance to the next)."""
    g = synth.make_random_graph(num_states=600, num_labels=40, mean_arcs=3.5, seed=3, final_frac=0.2)
    cfg = abi.decoder_config_recipe()
    cfg.beam, cfg.lattice_beam = 7.0, 4.0
    lls = make_set(g, 17, seed0=7, peaked=False)
    G = decoder.Graph(g)
    bd = decoder.BatchDecoder(G, cfg, abi.DecoderSizes(4, 1 << 14, 1 << 18, 1 << 19, 256))
    a, _, _ = bd.decode_queue(lls, resident_lanes=4)
    b, _, _ = bd.decode_queue(lls, resident_lanes=2, order=list(range(len(lls))))
    c, _, _ = bd.decode_queue(lls, resident_lanes=3, order=list(reversed(range(len(lls)))))
    for i, ll in enumerate(lls):
        lo = oracle_lattice(g, cfg, ll).GetRawLattice()
        assert lattices_equal(a[i], lo), "utt %d: %s" % (i, lattice_diff(a[i], lo) if a[i] is not None and lo is not None else (a[i], lo))
        assert lattices_equal(b[i], lo) and lattices_equal(c[i], lo)


def test_queue_after_batch_mode_and_back():
    """kamd_decoder_reserve splits the pools by utterance length for a batch launch; the queue
    needs the uniform split back, and a later batch launch must still work."""
    g = synth.make_hclg(num_units=24, vocab=90, n_hist=12, seed=9)
    cfg = abi.decoder_config_recipe()
    lls = make_set(g, 6, seed0=40)
    G = decoder.Graph(g)
    bd = decoder.BatchDecoder(G, cfg, abi.DecoderSizes(6, 1 << 13, 1 << 17, 1 << 18, 256))
    fr = np.asarray([ll.shape[0] for ll in lls], np.int32)
    assert lib().kamd_decoder_reserve(bd._dec, abi.iptr(fr), len(lls)) == 0
    ref = bd.decode(lls)
    q, _, _ = bd.decode_queue(lls, resident_lanes=2)
    again = bd.decode(lls)
    for i in range(len(lls)):
        assert lattices_equal(q[i], ref[i]) and lattices_equal(again[i], ref[i])


def test_queue_overflow_is_per_utterance():
    """One utterance that does not fit its lane's arena fails alone; the lane is clean for the next."""
    g = synth.make_hclg(num_units=40, vocab=150, n_hist=25, seed=5)
    cfg = abi.decoder_config_recipe()
    small = make_set(g, 6, seed0=300)
    big = synth.random_loglikes(200, g.num_pdfs, seed=1, scale=0.3)      # flat scores: many tokens per frame
    lls = small[:3] + [big] + small[3:]
    G = decoder.Graph(g)
    ntok = sum(oracle_lattice(g, cfg, ll).counters()[5] for ll in small)
    cap = int(max(oracle_lattice(g, cfg, ll).counters()[5] for ll in small)) * 2 + 64
    assert oracle_lattice(g, cfg, big).counters()[5] > 2 * cap and ntok > 0
    bd = decoder.BatchDecoder(G, cfg, abi.DecoderSizes(2, 1 << 14, cap, 8 * cap, 512))
    n = len(lls)
    dms = [decoder.DeviceMatrix(m) for m in lls]
    tasks = (abi.QueueTask * n)()
    for k in range(n):
        tasks[k] = abi.QueueTask(dms[k].ptr(0), dms[k].cols, dms[k].rows, k, 0)
    assert lib().kamd_decoder_queue_launch(bd._dec, tasks, n, 1, None) == 0       # ONE lane: everything after `big` reuses it
    ms, ln = C.c_float(), C.c_int32()
    assert lib().kamd_decoder_queue_wait(bd._dec, C.byref(ms), C.byref(ln)) == 0 and ln.value == 1
    for i, ll in enumerate(lls):
        r = abi.QueueResult()
        assert lib().kamd_decoder_queue_result(bd._dec, i, C.byref(r)) == 0
        if i == 3:
            assert r.error != 0                      # token arena (a later flag of the same frame may overwrite it)
            with pytest.raises(KamdError, match="capacity"):
                decoder.queue_fetch_lattice(bd._dec, i)
        else:
            assert r.error == 0
            assert lattices_equal(decoder.queue_fetch_lattice(bd._dec, i), oracle_lattice(g, cfg, ll).GetRawLattice())


def test_queue_argument_errors():
    g = synth.make_hclg(num_units=16, vocab=40, n_hist=6, seed=2)
    G = decoder.Graph(g)
    bd = decoder.BatchDecoder(G, abi.decoder_config_recipe(), abi.DecoderSizes(2, 1 << 12, 1 << 15, 1 << 16, 64))
    dm = decoder.DeviceMatrix(synth.random_loglikes(10, g.num_pdfs, seed=0))
    t = (abi.QueueTask * 1)(abi.QueueTask(dm.ptr(0), dm.cols, 100, 0, 0))       # more frames than max_frames
    assert lib().kamd_decoder_queue_launch(bd._dec, t, 1, 1, None) < 0
    t = (abi.QueueTask * 1)(abi.QueueTask(dm.ptr(0), dm.cols, 10, 5, 0))        # utterance index outside the table
    assert lib().kamd_decoder_queue_launch(bd._dec, t, 1, 1, None) < 0
    assert lib().kamd_decoder_queue_launch(bd._dec, t, 0, 1, None) < 0
