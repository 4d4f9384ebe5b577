"""final.mdl reader (kaldi_amd/mdl.py) against files written by tests/mdl_writer.py, which follows
the reference's Write functions.  PARITY UNPINNED: no model file exists in the reference tree."""
import numpy as np

from kaldi_amd import mdl, nnet
from oracle import orc
from tests.mdl_writer import write_mdl


def test_tdnnf_model_round_trip(tmp_path):
    m = nnet.make_tdnnf(48, 16, [1, 1, 0, 3, 3], 24, 60, input_dim=40, ivector_dim=10, seed=4)
    p = tmp_path / "final.mdl"
    id2pdf, tid_phone = write_mdl(p, m, num_units=30)
    got, id2pdf2, tid_phone2 = mdl.read_mdl(p, acoustic_scale=m.layers[-1].post_scale)
    np.testing.assert_array_equal(id2pdf, id2pdf2)
    np.testing.assert_array_equal(tid_phone, tid_phone2)
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
