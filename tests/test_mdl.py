"""final.mdl reader (kaldi_amd/mdl.py) against files written by tests/mdl_writer.py, which follows
the reference's Write functions.  PARITY UNPINNED: no model file exists in the reference tree."""
import numpy as np

from kaldi_amd import mdl, nnet
from oracle import orc
from tests.mdl_writer import write_mdl


def same_as_native(path, got, id2pdf, tid_phone, acoustic_scale=1.0, sub=3):
    """kamd_model_read (csrc/mdl.cc, the reader a C / C++ host uses) against the Python reader: every field bit for bit."""
    nat, id2pdf_n, tid_phone_n = mdl.read_mdl_native(path, acoustic_scale=acoustic_scale, frame_subsampling_factor=sub)
    np.testing.assert_array_equal(id2pdf, id2pdf_n)
    np.testing.assert_array_equal(tid_phone, tid_phone_n)
    np.testing.assert_array_equal(got.tid2phone, nat.tid2phone)
    assert (nat.input_dim, nat.ivector_dim, nat.subsampling, nat.num_pdfs, len(nat.layers)) == \
           (got.input_dim, got.ivector_dim, got.subsampling, got.num_pdfs, len(got.layers))
    for a, b in zip(nat.layers, got.layers):
        assert (a.in_dim, a.out_dim, list(a.offsets), a.input_layer, a.bypass_layer, bool(a.relu), a.ivector_dim, bool(a.log_softmax)) == \
               (b.in_dim, b.out_dim, list(b.offsets), b.input_layer, b.bypass_layer, bool(b.relu), b.ivector_dim, bool(b.log_softmax)), b.name
        assert np.float32(a.bypass_scale) == np.float32(b.bypass_scale) and np.float32(a.post_scale) == np.float32(b.post_scale), b.name
        assert a.slice_layers == (None if b.slice_layers is None else [int(q) for q in b.slice_layers]) and a.slice_dims == (None if b.slice_dims is None else [int(q) for q in b.slice_dims]), b.name
        for f in ("W", "bias", "bn_scale", "bn_offset", "post_offset"):
            x, y = getattr(a, f), getattr(b, f)
            assert (x is None) == (y is None), (b.name, f)
            if x is not None:
                np.testing.assert_array_equal(x, np.asarray(y, np.float32).reshape(x.shape), err_msg="%s.%s" % (b.name, f))


def test_tdnnf_model_round_trip(tmp_path):
    m = nnet.make_tdnnf(48, 16, [1, 1, 0, 3, 3], 24, 60, input_dim=40, ivector_dim=10, seed=4)
    p = tmp_path / "final.mdl"
    id2pdf, tid_phone = write_mdl(p, m, num_units=30)
    got, id2pdf2, tid_phone2 = mdl.read_mdl(p, acoustic_scale=m.layers[-1].post_scale)
    np.testing.assert_array_equal(id2pdf, id2pdf2)
    np.testing.assert_array_equal(tid_phone, tid_phone2)
    same_as_native(p, got, id2pdf2, tid_phone2, acoustic_scale=m.layers[-1].post_scale)
    assert len(got.layers) == len(m.layers) and got.input_dim == 40 and got.ivector_dim == 10
    for a, b in zip(got.layers, m.layers):
        assert (a.in_dim, a.out_dim, list(a.offsets), a.input_layer, a.bypass_layer, a.relu, a.ivector_dim) == \
               (b.in_dim, b.out_dim, list(b.offsets), b.input_layer, b.bypass_layer, b.relu, b.ivector_dim)
        np.testing.assert_array_equal(a.W, b.W)
        assert (a.bias is None) == (b.bias is None or not np.any(b.bias)) or np.array_equal(a.bias, b.bias)
        if b.bn_scale is not None:
            np.testing.assert_allclose(a.bn_scale, b.bn_scale, rtol=2e-6)
            np.testing.assert_allclose(a.bn_offset, b.bn_offset, rtol=2e-6, atol=1e-6)
        assert abs(a.bypass_scale - b.bypass_scale) < 1e-6
    rng = np.random.default_rng(0)
    feats = rng.standard_normal((70, 40)).astype(np.float32)
    iv = rng.standard_normal(10).astype(np.float32)
    ref = orc.nnet_forward(m, feats, iv)
    out = orc.nnet_forward(got, feats, iv)
    assert np.abs(out - ref).max() < 1e-4 * np.abs(ref).max()


def test_descriptor_parser():
    d = mdl._parse_descriptor("Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))")
    assert d == ("Append", [("Offset", ("node", "input"), -1), ("node", "input"), ("Offset", ("node", "input"), 1),
                            ("ReplaceIndex", "ivector")])
    d = mdl._parse_descriptor("Sum(Scale(0.66, tdnnf2.noop), tdnnf3.dropout)")
    assert d == ("Sum", ("Scale", 0.66, ("node", "tdnnf2.noop")), ("node", "tdnnf3.dropout"))


def test_transition_model_three_state_hmm_topology_old_format():
    """The GMM recipes' topology (3 emitting states, left to right, one pdf class per state: HmmTopology writes the
    non-extended form, and the tuples are <Triples>): id2pdf / phone tables worked out by hand."""
    from kaldi_amd import mdl
    from tests.mdl_writer import f32, i32, int_vector, tok, vec
    b = tok("<TransitionModel>") + tok("<Topology>")
    b += int_vector(np.array([1, 2])) + int_vector(np.array([-1, 0, 0]))         # phones, phone2idx
    b += i32(1)                                                                 # one topology entry, old format (no -1 marker)
    b += i32(4)                                                                 # four states
    for st in range(3):
        b += i32(st) + i32(2) + i32(st) + f32(0.75) + i32(st + 1) + f32(0.25)   # pdf class, two transitions: self, next
    b += i32(-1) + i32(0)                                                       # final state: no pdf class, no transitions
    b += tok("</Topology>") + tok("<Triples>") + i32(6)
    pdf = 0
    for ph in (1, 2):
        for hs in range(3):
            b += i32(ph) + i32(hs) + i32(pdf)
            pdf += 1
    b += tok("</Triples>") + tok("<LogProbs>") + vec(np.zeros(13, np.float32)) + tok("</LogProbs>") + tok("</TransitionModel>")
    s = mdl._Stream(b)
    id2pdf, tid_phone, phones = mdl.read_transition_model(s)
    assert list(phones) == [1, 2]
    # transition-ids in tuple order, two per state (self-loop, forward): both map to the state's pdf
    assert id2pdf.tolist() == [-1, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]
    # a phone is entered by the forward transition out of ... no: by any non-self-loop transition OF hmm-state 0
    assert tid_phone.tolist() == [0, 0, 1, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0]
    assert mdl.read_transition_model.tid2phone.tolist() == [0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2]


def test_components_outside_the_tdnnf_recipes(tmp_path):
    """The graph compiler is descriptor-driven, not name-driven: per-element components (FixedScale, FixedBias,
    PerElementScale, PerElementOffset with a block-repeated vector, ScaleAndOffset), a plain residual Sum(x, y), Scale(...)
    descriptors feeding an affine (whole input and one appended slice) and a lone Offset(...) input all fold into fused
    layers.  Checked against a direct float64 evaluation of the component chain as the reference's Propagate functions
    define it (nnet-simple-component.cc:1234, 2021, 2191-2225, 2400-2440, 3669, 3740, 957)."""
    from tests import mdl_writer as mw
    from tests.nnet_ref import forward_f64 as forward_ref
    rng = np.random.default_rng(5)
    D, H, P = 40, 24, 19
    W1, b1 = rng.standard_normal((H, 3 * D)) * 0.1, rng.standard_normal(H) * 0.1
    s1 = rng.uniform(0.5, 1.5, H)
    o1 = rng.standard_normal(8) * 0.1                      # block-repeated: 8 values over 24 dims
    W2, b2 = rng.standard_normal((H, H)) * 0.2, rng.standard_normal(H) * 0.1
    sc2, of2 = rng.uniform(0.5, 1.5, H), rng.standard_normal(H) * 0.1
    W3 = rng.standard_normal((H, 2 * H)) * 0.2             # LinearComponent over Append(Scale(2, Offset(l2,-1)), l2)
    pe3 = rng.uniform(0.5, 1.5, H)
    W4, b4 = rng.standard_normal((P, H)) * 0.2, rng.standard_normal(P) * 0.1
    fb4 = rng.standard_normal(P) * 0.1

    def affine(name, W, b):
        return (name, mw.updatable_common("NaturalGradientAffineComponent") + mw.tok("<LinearParams>") + mw.mat(W) + mw.tok("<BiasParams>") +
                mw.vec(b) + mw.tok("<RankIn>") + mw.i32(20) + mw.tok("<RankOut>") + mw.i32(80) + mw.tok("<UpdatePeriod>") + mw.i32(4) +
                mw.tok("<NumSamplesHistory>") + mw.f32(2000.0) + mw.tok("<Alpha>") + mw.f32(4.0) + mw.tok("</NaturalGradientAffineComponent>"))

    def relu(name, dim):
        z = np.zeros(dim, np.float32)
        return (name, mw.tok("<RectifiedLinearComponent>") + mw.tok("<Dim>") + mw.i32(dim) + mw.tok("<ValueAvg>") + mw.vec(z) +
                mw.tok("<DerivAvg>") + mw.vec(z) + mw.tok("<Count>") + mw.f64(0.0) + mw.tok("<OderivRms>") + mw.vec(z) +
                mw.tok("<OderivCount>") + mw.f64(0.0) + mw.tok("<NumDimsSelfRepaired>") + mw.f64(0.0) + mw.tok("<NumDimsProcessed>") +
                mw.f64(0.0) + mw.tok("<SelfRepairScale>") + mw.f32(1e-5) + mw.tok("</RectifiedLinearComponent>"))
    comps = [
        affine("l1.affine", W1, b1),
        ("l1.scale", mw.tok("<FixedScaleComponent>") + mw.tok("<Scales>") + mw.vec(s1) + mw.tok("</FixedScaleComponent>")),
        relu("l1.relu", H),
        ("l1.offset", mw.updatable_common("PerElementOffsetComponent") + mw.tok("<Offsets>") + mw.vec(o1) + mw.tok("<Dim>") + mw.i32(H) +
         mw.tok("<UseNaturalGradient>") + mw.boolean(True) + mw.tok("</PerElementOffsetComponent>")),
        affine("l2.affine", W2, b2),
        ("l2.so", mw.updatable_common("ScaleAndOffsetComponent") + mw.tok("<Dim>") + mw.i32(H) + mw.tok("<Scales>") + mw.vec(sc2) +
         mw.tok("<Offsets>") + mw.vec(of2) + mw.tok("<UseNaturalGradient>") + mw.boolean(True) + mw.tok("<Rank>") + mw.i32(20) +
         mw.tok("</ScaleAndOffsetComponent>")),
        relu("l2.relu", H),
        ("l2.res", mw.tok("<NoOpComponent>") + mw.tok("<Dim>") + mw.i32(H) + mw.tok("<BackpropScale>") + mw.f32(1.0) + mw.tok("</NoOpComponent>")),
        ("l3.linear", mw.updatable_common("LinearComponent") + mw.tok("<Params>") + mw.mat(W3) + mw.tok("<OrthonormalConstraint>") + mw.f32(-1.0) +
         mw.tok("<UseNaturalGradient>") + mw.boolean(True) + mw.tok("<RankInOut>") + mw.i32(20) + mw.i32(80) + mw.tok("<Alpha>") + mw.f32(4.0) +
         mw.tok("<NumSamplesHistory>") + mw.f32(2000.0) + mw.tok("<UpdatePeriod>") + mw.i32(4) + mw.tok("</LinearComponent>")),
        ("l3.pes", mw.updatable_common("PerElementScaleComponent") + mw.tok("<Params>") + mw.vec(pe3) + mw.tok("</PerElementScaleComponent>")),
        affine("output.affine", W4, b4),
        ("output.bias", mw.tok("<FixedBiasComponent>") + mw.tok("<Bias>") + mw.vec(fb4) + mw.tok("</FixedBiasComponent>")),
        ("output.log-softmax", mw.tok("<LogSoftmaxComponent>") + mw.tok("<Dim>") + mw.i32(P) + mw.tok("<ValueAvg>") + mw.vec(np.zeros(0)) +
         mw.tok("<DerivAvg>") + mw.vec(np.zeros(0)) + mw.tok("<Count>") + mw.f64(0.0) + mw.tok("<NumDimsSelfRepaired>") + mw.f64(0.0) +
         mw.tok("<NumDimsProcessed>") + mw.f64(0.0) + mw.tok("</LogSoftmaxComponent>")),
    ]
    cfg = ["input-node name=input dim=%d" % D,
           "component-node name=l1.affine component=l1.affine input=Append(Offset(input, -1), input, Offset(input, 1))",
           "component-node name=l1.scale component=l1.scale input=l1.affine",
           "component-node name=l1.relu component=l1.relu input=l1.scale",
           "component-node name=l1.offset component=l1.offset input=l1.relu",
           "component-node name=l2.affine component=l2.affine input=Scale(0.5, l1.offset)",
           "component-node name=l2.so component=l2.so input=l2.affine",
           "component-node name=l2.relu component=l2.relu input=l2.so",
           "component-node name=l2.res component=l2.res input=Sum(l1.offset, l2.relu)",
           "component-node name=l3.linear component=l3.linear input=Append(Scale(2.0, Offset(l2.res, -1)), l2.res)",
           "component-node name=l3.pes component=l3.pes input=l3.linear",
           "component-node name=output.affine component=output.affine input=Offset(l3.pes, 1)",
           "component-node name=output.bias component=output.bias input=output.affine",
           "component-node name=output.log-softmax component=output.log-softmax input=output.bias",
           "output-node name=output input=output.log-softmax objective=linear"]
    tm, id2pdf, _ = mw.transition_model(10)
    blob = b"\0B" + tm + mw.tok("<Nnet3>") + b"\n" + ("\n".join(cfg) + "\n\n").encode() + mw.tok("<NumComponents>") + mw.i32(len(comps))
    for name, body in comps:
        blob += mw.tok("<ComponentName>") + mw.tok(name) + body
    blob += mw.tok("</Nnet3>") + mw.tok("<LeftContext>") + mw.i32(0) + mw.tok("<RightContext>") + mw.i32(0) + mw.tok("<Priors>") + mw.vec(np.zeros(0))
    (tmp_path / "odd.mdl").write_bytes(blob)
    model, i2p, tph = mdl.read_mdl(tmp_path / "odd.mdl", acoustic_scale=1.0, frame_subsampling_factor=1)
    same_as_native(tmp_path / "odd.mdl", model, i2p, tph, sub=1)
    assert len(model.layers) == 4 and model.layers[1].bypass_layer == 0 and model.layers[1].bypass_scale == 1.0
    T = 23
    x = rng.standard_normal((T, D)).astype(np.float32)

    def at(m, t):                                           # nnet3 evaluates out-of-range times by clamping the INPUT (DecodableNnetSimple)
        return m[np.clip(t, 0, m.shape[0] - 1)]
    # direct evaluation over an extended time range so that only the network input is clamped, as the decodable does
    lo, hi = -4, T + 4
    ts = np.arange(lo, hi)
    xin = np.stack([x[np.clip(t, 0, T - 1)] for t in ts]).astype(np.float64)

    def shift(m, d):                                        # m[t + d] on the extended range (edges of the extension are never read back)
        return np.roll(m, -d, axis=0)
    l1 = np.concatenate([shift(xin, -1), xin, shift(xin, 1)], 1) @ W1.T + b1
    l1 = np.maximum(l1 * s1, 0.0) + np.tile(o1, H // 8)
    l2 = (0.5 * l1) @ W2.T + b2
    l2 = np.maximum(l2 * sc2 + of2, 0.0)
    res = l1 + l2
    l3 = (np.concatenate([2.0 * shift(res, -1), res], 1) @ W3.T) * pe3
    out = shift(l3, 1) @ W4.T + b4 + fb4
    out = out - np.log(np.exp(out - out.max(1, keepdims=True)).sum(1, keepdims=True)) - out.max(1, keepdims=True)
    want = out[-lo:-lo + T]
    got = forward_ref(model, x)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)


def test_native_reader_errors_name_the_problem(tmp_path):
    import pytest
    with pytest.raises(mdl.MdlError, match="cannot open"):
        mdl.read_mdl_native(tmp_path / "absent.mdl")
    (tmp_path / "text.mdl").write_bytes(b"<TransitionModel> \n")
    with pytest.raises(mdl.MdlError, match="binary Kaldi file expected"):
        mdl.read_mdl_native(tmp_path / "text.mdl")
    m = nnet.make_tdnnf(48, 16, [1, 0], 24, 60, input_dim=40, seed=4)
    p = tmp_path / "final.mdl"
    write_mdl(p, m, num_units=30)
    blob = p.read_bytes()
    (tmp_path / "cut.mdl").write_bytes(blob[:len(blob) // 2])
    with pytest.raises(mdl.MdlError, match="unexpected end of file"):
        mdl.read_mdl_native(tmp_path / "cut.mdl")
    # a count that the file cannot hold is refused before anything is sized by it: the <Tuples> count patched to 2^31 - 1
    at = blob.index(b"<Tuples> ") if b"<Tuples> " in blob else blob.index(b"<Triples> ")
    at = blob.index(b" ", at) + 1
    assert blob[at] == 4
    (tmp_path / "count.mdl").write_bytes(blob[:at + 1] + (2 ** 31 - 1).to_bytes(4, "little") + blob[at + 5:])
    with pytest.raises(mdl.MdlError, match="tuple count exceeds"):
        mdl.read_mdl_native(tmp_path / "count.mdl")
    bad = blob.replace(b"<RectifiedLinearComponent>", b"<SigmoidBlahBlahComponent>").replace(b"</RectifiedLinearComponent>", b"</SigmoidBlahBlahComponent>")
    assert bad != blob
    (tmp_path / "bad.mdl").write_bytes(bad)
    with pytest.raises(mdl.MdlError, match="unsupported component type SigmoidBlahBlahComponent"):
        mdl.read_mdl_native(tmp_path / "bad.mdl")
    with pytest.raises(mdl.MdlError, match="unsupported component type SigmoidBlahBlahComponent"):
        mdl.read_mdl(tmp_path / "bad.mdl")


def append_model(tmp_path, seed=11):
    """A small graph with an Append over DIFFERENT producers (and widths): l3 reads Append(Offset(l1, -1), l2, Offset(input, 2)),
    the output layer Append(l3, Scale(0.5, l1)).  -> (path, float64 reference forward as a function of the features)"""
    from tests import mdl_writer as mw
    rng = np.random.default_rng(seed)
    D, H1, H2, H3, P = 40, 24, 16, 32, 21
    W1, b1 = rng.standard_normal((H1, 2 * D)) * 0.1, rng.standard_normal(H1) * 0.1
    W2, b2 = rng.standard_normal((H2, H1)) * 0.2, rng.standard_normal(H2) * 0.1
    W3, b3 = rng.standard_normal((H3, H1 + H2 + D)) * 0.15, rng.standard_normal(H3) * 0.1
    W4, b4 = rng.standard_normal((P, H3 + H1)) * 0.2, rng.standard_normal(P) * 0.1
    W1, b1, W2, b2, W3, b3, W4, b4 = [a.astype(np.float32).astype(np.float64) for a in (W1, b1, W2, b2, W3, b3, W4, b4)]   # what the file holds

    def affine(name, W, b):
        return (name, mw.updatable_common("NaturalGradientAffineComponent") + mw.tok("<LinearParams>") + mw.mat(W) + mw.tok("<BiasParams>") +
                mw.vec(b) + mw.tok("<RankIn>") + mw.i32(20) + mw.tok("<RankOut>") + mw.i32(80) + mw.tok("<UpdatePeriod>") + mw.i32(4) +
                mw.tok("<NumSamplesHistory>") + mw.f32(2000.0) + mw.tok("<Alpha>") + mw.f32(4.0) + mw.tok("</NaturalGradientAffineComponent>"))

    def relu(name, dim):
        z = np.zeros(dim, np.float32)
        return (name, mw.tok("<RectifiedLinearComponent>") + mw.tok("<Dim>") + mw.i32(dim) + mw.tok("<ValueAvg>") + mw.vec(z) +
                mw.tok("<DerivAvg>") + mw.vec(z) + mw.tok("<Count>") + mw.f64(0.0) + mw.tok("<OderivRms>") + mw.vec(z) +
                mw.tok("<OderivCount>") + mw.f64(0.0) + mw.tok("<NumDimsSelfRepaired>") + mw.f64(0.0) + mw.tok("<NumDimsProcessed>") +
                mw.f64(0.0) + mw.tok("<SelfRepairScale>") + mw.f32(1e-5) + mw.tok("</RectifiedLinearComponent>"))
    comps = [affine("l1.affine", W1, b1), relu("l1.relu", H1), affine("l2.affine", W2, b2), relu("l2.relu", H2),
             affine("l3.affine", W3, b3), relu("l3.relu", H3), affine("output.affine", W4, b4)]
    cfg = ["input-node name=input dim=%d" % D,
           "component-node name=l1.affine component=l1.affine input=Append(Offset(input, -1), Offset(input, 1))",
           "component-node name=l1.relu component=l1.relu input=l1.affine",
           "component-node name=l2.affine component=l2.affine input=l1.relu",
           "component-node name=l2.relu component=l2.relu input=l2.affine",
           "component-node name=l3.affine component=l3.affine input=Append(Offset(l1.relu, -1), l2.relu, Offset(input, 2))",
           "component-node name=l3.relu component=l3.relu input=l3.affine",
           "component-node name=output.affine component=output.affine input=Append(l3.relu, Scale(0.5, l1.relu))",
           "output-node name=output input=output.affine objective=linear"]
    tm, _, _ = mw.transition_model(10)
    blob = b"\0B" + tm + mw.tok("<Nnet3>") + b"\n" + ("\n".join(cfg) + "\n\n").encode() + mw.tok("<NumComponents>") + mw.i32(len(comps))
    for name, body in comps:
        blob += mw.tok("<ComponentName>") + mw.tok(name) + body
    blob += mw.tok("</Nnet3>") + mw.tok("<LeftContext>") + mw.i32(0) + mw.tok("<RightContext>") + mw.i32(0) + mw.tok("<Priors>") + mw.vec(np.zeros(0))
    path = tmp_path / "append.mdl"
    path.write_bytes(blob)

    def direct(x):
        T = x.shape[0]
        lo, hi = -6, T + 6
        ts = np.arange(lo, hi)
        xin = np.stack([x[np.clip(t, 0, T - 1)] for t in ts]).astype(np.float64)
        sh = lambda m, d: np.roll(m, -d, axis=0)                     # m[t + d] on the extended range
        l1 = np.maximum(np.concatenate([sh(xin, -1), sh(xin, 1)], 1) @ W1.T + b1, 0.0)
        l2 = np.maximum(l1 @ W2.T + b2, 0.0)
        l3 = np.maximum(np.concatenate([sh(l1, -1), l2, sh(xin, 2)], 1) @ W3.T + b3, 0.0)
        out = np.concatenate([l3, 0.5 * l1], 1) @ W4.T + b4
        return out[-lo:-lo + T]
    return path, direct


def test_append_over_different_producers(tmp_path):
    """Descriptors like Append(Offset(l1, -1), l2, Offset(input, 2)) -- evaluated by the reference with kCopyRows over
    arbitrary sources (nnet3/nnet-compute.cc:309-383) -- compile into multi-input layers in both readers; the oracle's
    forward of that model equals a direct float64 evaluation of the graph."""
    from tests.nnet_ref import forward_f64
    path, direct = append_model(tmp_path)
    model, i2p, tph = mdl.read_mdl(path, acoustic_scale=1.0, frame_subsampling_factor=1)
    same_as_native(path, model, i2p, tph, sub=1)
    l3, out = model.layers[2], model.layers[3]
    assert l3.slice_layers == [0, 1, -1] and l3.slice_dims == [24, 16, 40] and list(l3.offsets) == [-1, 0, 2] and l3.in_dim == 80
    assert out.slice_layers == [2, 0] and out.slice_dims == [32, 24]
    assert model.context() == (2, 2)                  # l1: +-1; l3: l1 at -1 -> left 2; input at +2 -> right 2
    x = np.random.default_rng(2).standard_normal((29, 40)).astype(np.float32)
    want = direct(x)
    np.testing.assert_allclose(forward_f64(model, x), want, rtol=1e-9, atol=1e-9)
    got = orc.nnet_forward(model, x)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * np.abs(want).max())
