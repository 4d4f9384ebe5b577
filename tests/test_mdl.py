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
