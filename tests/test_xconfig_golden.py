"""The model topology pinned by the REFERENCE'S OWN xconfig generator.

tests/golden/nnet/tdnn_1d.final.config / tdnn_1h.final.config are what steps/libs/nnet3/xconfig (imported
from the reference tree by tools/gen_xconfig_golden.py in the build container) emits for the xconfigs of
egs/librispeech/s5/local/chain/tuning/run_tdnn_1d.sh:219-249 and
egs/mini_librispeech/s5/local/chain/tuning/run_tdnn_1h.sh:163-190.  Checked here:

 (i)  kaldi_amd/mdl.py compiles a final.mdl whose config section is that text (random parameters, written like
      the reference's Write functions) into exactly the fused-layer list nnet.tdnnf_librispeech() /
      tdnnf_mini_librispeech() build: dims, time offsets, producers, bypass producer and scale, ReLU / BatchNorm
      flags, i-vector input, model context;
 (ii) the forward of that model (CPU oracle here; the HIP path in tests/test_gpu_nnet.py::test_xconfig_models)
      equals a direct float64 evaluation of the component graph, node by node (tests/xconfig_mdl.py), which
      shares no code with mdl.py's compile.
"""
import os

import numpy as np
import pytest

from kaldi_amd import mdl, nnet
from oracle import orc
from tests import xconfig_mdl

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nnet")
CASES = {"tdnn_1d": (nnet.tdnnf_librispeech, 6000, 0.75), "tdnn_1h": (nnet.tdnnf_mini_librispeech, 2328, 0.66)}


def _load(name, tmp_path, seed=0):
    text = open(os.path.join(GOLD, name + ".final.config")).read()
    params = xconfig_mdl.random_params(text, seed=seed)
    P = CASES[name][1]
    rng = np.random.default_rng(seed + 1)
    priors = rng.dirichlet(np.full(P, 5.0)).astype(np.float32)
    path = tmp_path / (name + ".mdl")
    xconfig_mdl.write_mdl_from_config(path, text, params, priors, num_units=P // 2)
    model, id2pdf, tid_phone = mdl.read_mdl(path, acoustic_scale=1.0)
    from tests.test_mdl import same_as_native
    same_as_native(path, model, id2pdf, tid_phone)          # kamd_model_read (csrc/mdl.cc) reads the same file to the same bits
    return text, params, priors, model


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_layers_equal_the_bench_topology(name, tmp_path):
    make, P, bypass = CASES[name]
    text, params, priors, got = _load(name, tmp_path)
    want = make(num_pdfs=P, ivector_dim=100)
    assert got.input_dim == 40 and got.ivector_dim == 100 and got.subsampling == 3
    assert len(got.layers) == len(want.layers)
    for a, b in zip(got.layers, want.layers):
        assert (a.in_dim, a.out_dim, list(a.offsets), a.input_layer, a.bypass_layer, bool(a.relu), a.ivector_dim,
                a.bn_scale is not None, a.bias is not None, a.log_softmax) == \
               (b.in_dim, b.out_dim, list(b.offsets), b.input_layer, b.bypass_layer, bool(b.relu), b.ivector_dim,
                b.bn_scale is not None, b.bias is not None, b.log_softmax), (a.name, b.name)
        assert a.W.shape == b.W.shape
        if b.bypass_layer != -2:
            assert abs(a.bypass_scale - bypass) < 1e-7 and abs(b.bypass_scale - bypass) < 1e-7, a.name
    assert got.context() == want.context()
    # the context nnet3-info would print for these recipes (left 1+1+1+0+3*12 (+1 lda) ... computed from the config text)
    n_tdnnf3 = sum(1 for l in text.split("\n") if "time-offsets=-3,0" in l)
    n_tdnnf1 = sum(1 for l in text.split("\n") if "time-offsets=-1,0" in l)
    assert got.context() == (1 + n_tdnnf1 + 3 * n_tdnnf3, 1 + n_tdnnf1 + 3 * n_tdnnf3)
    assert abs(want.macs_per_output_frame() - got.macs_per_output_frame()) < 1


def test_forward_equals_direct_graph_evaluation(tmp_path):
    """tdnn_1h (mini_librispeech) at full width: oracle forward of the mdl.py model vs the node-by-node float64
    evaluation of the generated config.  (The 1d model's forward is compared on the device, -m gpu.)"""
    text, params, priors, model = _load("tdnn_1h", tmp_path, seed=3)
    rng = np.random.default_rng(11)
    T = 50
    feats = rng.standard_normal((T, 40)).astype(np.float32)
    iv = rng.standard_normal(100).astype(np.float32)
    want = xconfig_mdl.evaluate(text, params, feats, iv, priors=priors)
    got = orc.nnet_forward(model, feats, iv)
    assert got.shape == want.shape == ((T + 2) // 3, 2328)
    assert np.abs(got - want).max() < 1e-4 * np.abs(want).max()


def test_golden_text_is_what_the_recipe_says():
    """Facts of the recipe scripts that the generated text must carry (run_tdnn_1d.sh:209-249, run_tdnn_1h.sh:156-190)."""
    d = open(os.path.join(GOLD, "tdnn_1d.final.config")).read()
    h = open(os.path.join(GOLD, "tdnn_1h.final.config")).read()
    assert d.count("Sum(Scale(0.75, ") == 16 and "Scale(0.66" not in d
    assert h.count("Sum(Scale(0.66, ") == 12 and "Scale(0.75" not in h
    assert "input=Append(Offset(input, -1), input, Offset(input, 1), ReplaceIndex(ivector, t, 0))" in d
    assert "output-node name=output input=output.affine objective=linear" in d
    assert d.count("type=TdnnComponent") == 32 and h.count("type=TdnnComponent") == 24
    assert "input-dim=1536 output-dim=160" in d and "input-dim=768 output-dim=96" in h
    assert "LogSoftmaxComponent" not in d.split("prefinal-xent")[0]       # include-log-softmax=false on the chain output
