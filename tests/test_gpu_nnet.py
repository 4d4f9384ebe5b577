"""GPU parity: fused MFMA TDNN-F forward vs the CPU oracle.  Tolerance: 1e-4 of the output
scale (fp32 products, fp32 accumulation in a different order)."""
import numpy as np
import pytest

from kaldi_amd import decoder, nnet
from kaldi_amd._lib import KamdError
from oracle import orc

pytestmark = pytest.mark.gpu


def check(model, T, seed, ivec=None):
    rng = np.random.default_rng(seed)
    feats = (3 * rng.standard_normal((T, model.input_dim))).astype(np.float32)
    got = decoder.Nnet(model).Forward(feats, ivec)
    ref = orc.nnet_forward(model, feats, ivec)
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 1e-4 * scale, (np.abs(got - ref).max(), scale)
    return got


@pytest.mark.parametrize("T", [1, 2, 3, 50, 151, 400])
def test_tiny_model(T):
    check(nnet.tdnnf_tiny(num_pdfs=70), T, T)


def test_ivector_input():
    m = nnet.tdnnf_tiny(num_pdfs=33, ivector_dim=10)
    iv = np.random.default_rng(1).standard_normal(10).astype(np.float32)
    check(m, 77, 2, iv)


def test_odd_dims_padding():
    m = nnet.make_tdnnf(50, 18, [1, 0, 3], 22, 45, input_dim=13, seed=9)
    check(m, 64, 3)


@pytest.mark.parametrize("P", [33, 2328])
def test_log_softmax_output(P):
    m = nnet.tdnnf_tiny(num_pdfs=P, seed=5)
    m.layers[-1].log_softmax = True
    got = check(m, 91, 6)
    out = m.layers[-1]
    lp = got / out.post_scale - out.post_offset
    np.testing.assert_allclose(np.exp(lp.astype(np.float64)).sum(axis=1), 1.0, atol=1e-3)


def test_chunked_forward_with_online_ivectors():
    """kamd_nnet_forward_chunked_device == the oracle's DecodableNnetSimple with online ivectors,
    for a ragged batch; frames_per_chunk 50 (-> 51) and 21."""
    m = nnet.tdnnf_tiny(num_pdfs=41, ivector_dim=10, seed=7)
    N = decoder.Nnet(m)
    rng = np.random.default_rng(2)
    feats, ivs = [], []
    for T in (140, 17, 1, 263):
        feats.append((2 * rng.standard_normal((T, m.input_dim))).astype(np.float32))
        ivs.append(rng.standard_normal(((T + 9) // 10, 10)).astype(np.float32))
    for fpc in (50, 21):
        got = N.ForwardChunked(feats, ivs, 10, fpc)
        for f, iv, g in zip(feats, ivs, got):
            ref = orc.nnet_forward_chunked(m, f, iv, 10, fpc)
            assert g.shape == ref.shape
            assert np.abs(g - ref).max() < 1e-4 * np.abs(ref).max()
    # one ivector per utterance: chunked == plain forward
    const = [np.tile(iv[:1], (iv.shape[0], 1)) for iv in ivs]
    got = N.ForwardChunked(feats, const, 10, 50)
    for f, iv, g in zip(feats, const, got):
        plain = N.Forward(f, iv[0])
        assert np.abs(g - plain).max() < 1e-4 * np.abs(plain).max()


def test_batch_computer_tasks_with_online_ivectors():
    """kamd_nnet_forward_tasks_device == the oracle's NnetBatchComputer (SplitUtteranceIntoTasks + Compute + MergeTaskOutput,
    nnet3/nnet-batch-compute.cc:586-870) for a ragged batch: utterances shorter than a task, exactly one task, one frame
    more, several tasks with an overlapping last one; frames_per_chunk 50 (16 output frames) and 21 (7)."""
    m = nnet.tdnnf_tiny(num_pdfs=41, ivector_dim=10, seed=7)
    N = decoder.Nnet(m)
    rng = np.random.default_rng(4)
    feats, ivs = [], []
    for T in (140, 17, 1, 48, 49, 263):
        feats.append((2 * rng.standard_normal((T, m.input_dim))).astype(np.float32))
        ivs.append(rng.standard_normal(((T + 9) // 10, 10)).astype(np.float32))
    for fpc in (50, 21):
        got = N.ForwardChunked(feats, ivs, 10, fpc, batch_computer=True)
        for f, iv, g in zip(feats, ivs, got):
            ref = orc.nnet_forward_batch_computer(m, f, iv, 10, fpc)
            assert g.shape == ref.shape
            assert np.abs(g - ref).max() < 1e-4 * np.abs(ref).max()
    simple = N.ForwardChunked(feats, ivs, 10, 50)
    tasks = N.ForwardChunked(feats, ivs, 10, 50, batch_computer=True)
    assert np.abs(simple[0] - tasks[0]).max() > 1e-3          # two different chunkings of the same inputs


def test_a_minibatch_of_inference_tasks_in_any_order():
    """kamd_nnet_forward_inference_tasks_device: the unit NnetBatchComputer::Compute evaluates (nnet-batch-compute.cc:398-470), for a
    host that keeps the reference's scheduler.  The tasks the oracle's SplitUtteranceIntoTasks makes of three utterances are
    handed over in a shuffled order (a scheduler picks by priority, not by utterance), tasks of different shapes in ONE call;
    every task's used rows must be the oracle's NnetBatchComputer rows for that utterance.  A second call evaluates a model
    without the i-vector input (iv_row -1)."""
    m = nnet.tdnnf_tiny(num_pdfs=41, ivector_dim=10, seed=7)
    N = decoder.Nnet(m)
    rng = np.random.default_rng(9)
    feats, ivs, refs, tasks = [], [], [], []
    iv_base = 0
    for u, T in enumerate((140, 17, 263)):
        feats.append((2 * rng.standard_normal((T, m.input_dim))).astype(np.float32))
        ivs.append(rng.standard_normal(((T + 9) // 10, 10)).astype(np.float32))
        ref, tab = orc.nnet_forward_batch_computer(m, feats[u], ivs[u], 10, 50, return_tasks=True)
        refs.append(ref)
        for first_used, _, n_used, _, _, iv_row in tab.tolist():
            tasks.append((u, first_used, n_used, iv_base + iv_row))
        iv_base += ivs[u].shape[0]
    order = rng.permutation(len(tasks))
    shuffled = [tasks[i] for i in order]
    assert len({t[2] for t in shuffled}) > 1                    # tasks of different lengths in one minibatch
    outs = N.ForwardInferenceTasks(feats, np.concatenate(ivs), shuffled)
    for (u, first, n, _), got in zip(shuffled, outs):
        want = refs[u][first:first + n]
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4 * np.abs(refs[u]).max()
    with pytest.raises(Exception):                             # a task beyond the utterance's last output frame
        N.ForwardInferenceTasks(feats, np.concatenate(ivs), [(1, 4, 5, 0)])
    m0 = nnet.tdnnf_tiny(num_pdfs=41, seed=7)
    N0 = decoder.Nnet(m0)
    whole = N0.Forward(feats[0])
    got = N0.ForwardInferenceTasks(feats, None, [(0, 10, 7, -1), (0, 0, 3, -1)])
    assert np.abs(got[0] - whole[10:17]).max() < 1e-4 * np.abs(whole).max() and np.abs(got[1] - whole[0:3]).max() < 1e-4 * np.abs(whole).max()


def test_context_and_plan():
    m = nnet.tdnnf_mini_librispeech(num_pdfs=64)
    n = decoder.Nnet(m)
    assert n.Context() == m.context() == (28, 28)


def test_mini_librispeech_topology_small_T():
    m = nnet.tdnnf_mini_librispeech(num_pdfs=200)
    check(m, 40, 5)


def test_batch_invariance():
    """An utterance's output does not depend on what else is in the batch."""
    import ctypes as C
    from kaldi_amd import abi
    from kaldi_amd._lib import check as ck, lib
    m = nnet.tdnnf_tiny(num_pdfs=40)
    n = decoder.Nnet(m)
    rng = np.random.default_rng(0)
    Ts = [31, 90, 7]
    feats = [rng.standard_normal((T, 40)).astype(np.float32) for T in Ts]
    single = [n.Forward(f) for f in feats]
    ld = 48
    rows = sum(Ts)
    big = np.zeros((rows, ld), np.float32)
    off = np.concatenate([[0], np.cumsum(Ts)]).astype(np.int64)
    for f, o in zip(feats, off):
        big[o:o + f.shape[0], :40] = f
    nout = [(T + 2) // 3 for T in Ts]
    ooff = np.concatenate([[0], np.cumsum(nout)]).astype(np.int64)
    d_in = decoder.DeviceMatrix(big)
    d_out = lib().kamd_malloc(int(ooff[-1]) * 40 * 4)
    ck(lib().kamd_nnet_forward_batch_device(n._h, d_in.ptr(0), abi.iptr(off, C.c_int64), ld, None, 3,
                                            d_out, abi.iptr(ooff, C.c_int64), 40, None))
    ck(lib().kamd_device_synchronize())
    out = np.zeros((int(ooff[-1]), 40), np.float32)
    ck(lib().kamd_memcpy_d2h(out.ctypes.data_as(C.c_void_p), d_out, out.nbytes))
    lib().kamd_free(d_out)
    for u in range(3):
        np.testing.assert_array_equal(out[ooff[u]:ooff[u + 1]], single[u])


def test_pipeline_with_per_utterance_ivectors():
    """nnet3-latgen-faster --ivectors: one constant ivector per utterance through the batched
    pipeline == the single-utterance forward with that ivector == the oracle."""
    from kaldi_amd import abi, feat, pipeline, synth
    g = synth.make_hclg(num_units=20, vocab=30, n_hist=6, seed=1)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, ivector_dim=10, output_scale=2.0)
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, abi.decoder_config_recipe(), max_utts=3, max_seconds=3.0)
    waves = [synth.make_wave(s, seed=i) for i, s in enumerate((0.9, 1.7, 0.4))]
    rng = np.random.default_rng(4)
    ivs = rng.standard_normal((3, 10)).astype(np.float32)
    pipe.load(waves)
    pipe.set_ivectors(ivs)
    pipe.run()
    N = decoder.Nnet(m)
    for u in range(3):
        f = pipe.features(u)
        want = N.Forward(f, ivs[u])
        np.testing.assert_array_equal(pipe.loglikes(u), want)
        ref = orc.nnet_forward(m, f, ivs[u])
        assert np.abs(want - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_pipeline_with_online_ivectors():
    """wav -> MFCC -> chunked nnet with online ivectors -> decoder: the log-likelihoods the
    decoder consumed are the oracle's DecodableNnetSimple rows."""
    from kaldi_amd import abi, feat, pipeline, synth
    g = synth.make_hclg(num_units=20, vocab=40, n_hist=8, seed=2)
    m = nnet.tdnnf_tiny(num_pdfs=g.num_pdfs, ivector_dim=10, seed=1)
    cfg = abi.decoder_config_recipe()
    waves = [synth.make_wave(d, seed=30 + i) for i, d in enumerate((1.4, 0.6, 2.2))]
    pipe = pipeline.Pipeline(abi.mfcc_opts_hires(), m, g, cfg, max_utts=3, max_seconds=3.0)
    pipe.load(waves)
    rng = np.random.default_rng(5)
    mf = feat.Mfcc(abi.mfcc_opts_hires())
    feats = [mf.ComputeFeatures(w) for w in waves]
    ivs = [rng.standard_normal(((f.shape[0] + 9) // 10, 10)).astype(np.float32) for f in feats]
    pipe.set_online_ivectors(ivs, 10, 50)
    pipe.run()
    for u in range(3):
        ref = orc.nnet_forward_chunked(m, feats[u], ivs[u], 10, 50)
        got = pipe.loglikes(u)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()
    pipe.set_online_ivectors(None)
    with pytest.raises(Exception):
        pipe.run()                      # the model needs ivectors


def test_model_read_from_mdl_file(tmp_path):
    """final.mdl -> kaldi_amd.mdl.read_mdl -> device forward == the model it was written from"""
    from kaldi_amd import mdl
    from tests.mdl_writer import write_mdl
    m = nnet.make_tdnnf(64, 16, [1, 0, 3], 32, 50, input_dim=40, seed=11)
    write_mdl(tmp_path / "final.mdl", m, num_units=25)
    got, id2pdf, tid_phone = mdl.read_mdl(tmp_path / "final.mdl", acoustic_scale=m.layers[-1].post_scale)
    feats = (2 * np.random.default_rng(3).standard_normal((90, 40))).astype(np.float32)
    a = decoder.Nnet(got).Forward(feats)
    b = decoder.Nnet(m).Forward(feats)
    assert np.abs(a - b).max() < 1e-4 * np.abs(b).max()
    assert id2pdf.size == 51 and tid_phone.max() == 25


def test_looped_ivector_slots():
    """Round(ivector, period): first-layer rows read the i-vector slot of their own time.  Whole utterance against
    the oracle; slices in the middle of the utterance give the same rows as the whole (bit-equal), which is what
    lets the streaming path recompute context rows without changing them."""
    from kaldi_amd import decoder
    m = nnet.tdnnf_tiny(num_pdfs=40, ivector_dim=12)
    N = decoder.Nnet(m)
    rng = np.random.default_rng(4)
    T = 131
    x = rng.standard_normal((T, 40)).astype(np.float32)
    period = 20
    first = -2
    tab = rng.standard_normal((10, 12)).astype(np.float32)          # slots -2 .. 7 cover t in [-40, 160)
    whole = N.ForwardSlots(x, tab, first, period)[0]
    want = orc.nnet_forward_slots(m, x, tab, first, period)
    np.testing.assert_allclose(whole, want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))
    # a table whose slots are all equal is the per-utterance i-vector
    same = N.ForwardSlots(x, np.tile(tab[3], (10, 1)), first, period)[0]
    np.testing.assert_array_equal(same, N.Forward(x, ivector=tab[3]))
    L, R = N.Context()
    sub = m.subsampling
    o0, o1 = 9, 30                                                   # output frames [9, 30)
    k0 = min(o0, (L + sub - 1) // sub)
    in_first, in_last = sub * (o0 - k0), min(T - 1, sub * (o1 - 1) + R)
    part = N.ForwardSlots(x, tab, first, period, slices=[(in_first, in_last - in_first + 1), (0, T)])
    np.testing.assert_array_equal(part[0][k0:k0 + (o1 - o0)], whole[o0:o1])
    np.testing.assert_array_equal(part[1], whole)


@pytest.mark.parametrize("name", ["tdnn_1d", "tdnn_1h"])
def test_xconfig_models(name, tmp_path):
    """The recipe models as the REFERENCE'S OWN xconfig generator writes them (tests/golden/nnet/*.final.config, see
    tools/gen_xconfig_golden.py), full width, random parameters, read back through kaldi_amd/mdl.py: the device forward
    == the CPU oracle == a direct float64 evaluation of the component graph node by node (tests/xconfig_mdl.py)."""
    from tests import xconfig_mdl
    from tests.test_xconfig_golden import _load
    text, params, priors, model = _load(name, tmp_path, seed=5)
    rng = np.random.default_rng(17)
    iv = rng.standard_normal(100).astype(np.float32)
    for T in (40, 133):
        feats = rng.standard_normal((T, 40)).astype(np.float32)
        got = decoder.Nnet(model).Forward(feats, iv)
        want = xconfig_mdl.evaluate(text, params, feats, iv, priors=priors)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-4 * np.abs(want).max()
        if T == 40:
            ref = orc.nnet_forward(model, feats, iv)
            assert np.abs(got - ref).max() < 1e-4 * np.abs(ref).max()


def test_component_propagate_shaped_entry():
    """kamd_component_propagate = nnet3::Component::Propagate for one fused layer (TdnnComponent::Propagate's shape,
    nnet-tdnn-component.cc:181-212: consecutive time steps in, in_rows - span rows out), against the formula in float64:
    a TDNN layer with three offsets + bias + ReLU + BatchNorm map, a LinearComponent (no bias), an affine with one offset."""
    from kaldi_amd.nnet import Layer
    rng = np.random.default_rng(3)
    for in_dim, out_dim, offs, bias, relu, bn in ((40, 96, [-1, 0, 2], True, True, True), (96, 24, [-3, 0], False, False, False),
                                                  (24, 150, [0], True, False, False)):
        W = (rng.standard_normal((out_dim, len(offs) * in_dim)) / np.sqrt(len(offs) * in_dim)).astype(np.float32)
        b = rng.standard_normal(out_dim).astype(np.float32) * 0.1 if bias else None
        sc = rng.uniform(0.5, 1.5, out_dim).astype(np.float32) if bn else None
        of = rng.standard_normal(out_dim).astype(np.float32) * 0.1 if bn else None
        comp = decoder.Component(Layer("c", in_dim, out_dim, offs, -1, W, b, relu, sc, of))
        for T in (1, 7, 133):
            x = rng.standard_normal((T, in_dim)).astype(np.float32)
            got = comp.Propagate(x)
            span = max(offs) - min(offs)
            assert got.shape == (max(0, T - span), out_dim)
            if T <= span:
                continue
            rows = np.arange(T - span)
            want = sum(x[rows + o - min(offs)].astype(np.float64) @ W[:, k * in_dim:(k + 1) * in_dim].T.astype(np.float64)
                       for k, o in enumerate(offs))
            if bias:
                want = want + b
            if relu:
                want = np.maximum(want, 0.0)
            if bn:
                want = want * sc + of
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
    with pytest.raises(KamdError, match="offsets must include"):
        decoder.Component(Layer("c", 8, 8, [1, 2], -1, np.zeros((8, 16), np.float32)))


def test_append_over_different_producers_on_the_device(tmp_path):
    """A graph with Append over different producers and widths (tests/test_mdl.py::append_model; the reference: kCopyRows
    over arbitrary sources, nnet3/nnet-compute.cc:309-383): the device materialises the Append (a concat pseudo-layer in
    front of the GEMM) -- whole utterances, a batch with ragged lengths, and subsampled outputs, against the oracle and a
    direct float64 evaluation of the graph."""
    from kaldi_amd import mdl
    from tests.test_mdl import append_model
    path, direct = append_model(tmp_path)
    rng = np.random.default_rng(5)
    for sub in (1, 3):
        model, _, _ = mdl.read_mdl(path, acoustic_scale=1.0, frame_subsampling_factor=sub)
        N = decoder.Nnet(model)
        assert N.Context() == (2, 2)
        for T in (1, 2, 7, 64, 301):
            x = rng.standard_normal((T, 40)).astype(np.float32)
            got = N.Forward(x)
            want = direct(x)[::sub]
            assert got.shape == want.shape
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-5 * max(1.0, np.abs(want).max()))
            np.testing.assert_allclose(got, orc.nnet_forward(model, x), rtol=0, atol=3e-5 * max(1.0, np.abs(want).max()))
