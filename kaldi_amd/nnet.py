"""TDNN(-F) chain acoustic model description (the collapsed nnet3 inference graph).

Mirrors what the reference holds after `CollapseModel` (nnet3/nnet-utils.cc:2006) for
the topologies of egs/librispeech/s5/local/chain/tuning/run_tdnn_1d.sh:219-249 and
egs/mini_librispeech/s5/local/chain/tuning/run_tdnn_1h.sh:163-190: a chain of fused
layers, each = TdnnComponent/Affine/Linear + ReLU + test-mode BatchNorm + tdnnf bypass.
Weights are synthetic (random init): there is no network access for real models.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi


@dataclass
class Layer:
    name: str
    in_dim: int
    out_dim: int
    offsets: List[int]
    input_layer: int              # -1 = network input
    W: np.ndarray                 # [out_dim, n_offsets*in_dim + ivector_dim] float32
    bias: Optional[np.ndarray] = None
    relu: bool = False
    bn_scale: Optional[np.ndarray] = None
    bn_offset: Optional[np.ndarray] = None
    bypass_layer: int = -2        # -2 none
    bypass_scale: float = 0.0
    ivector_dim: int = 0
    post_offset: Optional[np.ndarray] = None
    post_scale: float = 1.0
    log_softmax: bool = False     # LogSoftmaxComponent on the output node (non-chain models)
    # Append over DIFFERENT producers (kamd_layer_desc::multi_input): slice j reads layer slice_layers[j] (-1 = input)
    # at offsets[j] and is slice_dims[j] wide; in_dim = sum(slice_dims), input_layer is ignored
    slice_layers: Optional[List[int]] = None
    slice_dims: Optional[List[int]] = None

    def slices(self):
        """[(producer, offset, first column of W, width)] of the layer's input."""
        if self.slice_layers is None:
            return [(self.input_layer, o, j * self.in_dim, self.in_dim) for j, o in enumerate(self.offsets)]
        col = np.concatenate([[0], np.cumsum(self.slice_dims)]).astype(int)
        return [(self.slice_layers[j], o, int(col[j]), int(self.slice_dims[j])) for j, o in enumerate(self.offsets)]


@dataclass
class Model:
    layers: List[Layer]
    input_dim: int
    ivector_dim: int
    subsampling: int = 3
    num_pdfs: int = 0
    name: str = ""
    _keep: list = field(default_factory=list, repr=False)

    def descs(self):
        """ctypes array of kamd_layer_desc (keeps the numpy buffers alive)."""
        arr = (abi.LayerDesc * len(self.layers))()
        for i, l in enumerate(self.layers):
            d = arr[i]
            d.in_dim, d.out_dim, d.n_offsets = l.in_dim, l.out_dim, len(l.offsets)
            for j, o in enumerate(l.offsets):
                d.offsets[j] = o
            d.input_layer, d.ivector_dim = l.input_layer, l.ivector_dim
            d.bypass_layer, d.bypass_scale, d.relu = l.bypass_layer, l.bypass_scale, int(l.relu)
            for nm in ("W", "bias", "bn_scale", "bn_offset", "post_offset"):
                a = getattr(l, nm)
                if a is not None:
                    a = np.ascontiguousarray(a, dtype=np.float32)
                    setattr(l, nm, a)
                setattr(d, nm, abi.fptr(a))
            d.post_scale = l.post_scale
            d.log_softmax = int(l.log_softmax)
            d.multi_input = 0
            if l.slice_layers is not None:
                d.multi_input = 1
                for j in range(len(l.offsets)):
                    d.slice_layer[j], d.slice_dim[j] = int(l.slice_layers[j]), int(l.slice_dims[j])
        self._keep.append(arr)
        return arr

    def context(self):
        """ComputeSimpleNnetContext (nnet3/nnet-utils.cc:146): (left, right)."""
        memo = {}

        def ctx(i):
            if i == -1:
                return 0, 0
            if i not in memo:
                memo[i] = ctx1(i)
            return memo[i]

        def ctx1(i):
            l = self.layers[i]
            left = right = 0
            for prod, off, _, _ in l.slices():
                il, ir = ctx(prod)
                left, right = max(left, il - min(0, off)), max(right, ir + max(0, off))
            if l.bypass_layer != -2:
                bl, br = ctx(l.bypass_layer)
                left, right = max(left, bl), max(right, br)
            return left, right
        return ctx(len(self.layers) - 1)

    def macs_per_output_frame(self):
        """Algorithmic MACs per output frame (each layer evaluated at
        the rate its consumers need, no chunk-edge recomputation)."""
        n = len(self.layers)
        # needed time residues: rate = fraction of 100fps frames needed.
        need = [set() for _ in range(n + 1)]  # index n = input (-1)
        need[n - 1] = {0}
        period = self.subsampling
        for i in range(n - 1, -1, -1):
            l = self.layers[i]
            for prod, o, _, _ in l.slices():
                tgt = need[prod] if prod >= 0 else need[n]
                for r in need[i]:
                    tgt.add((r + o) % period)
            if l.bypass_layer != -2:
                b = need[l.bypass_layer] if l.bypass_layer >= 0 else need[n]
                b.update(need[i])
        macs = 0.0
        for i, l in enumerate(self.layers):
            macs += len(need[i]) * l.W.shape[0] * l.W.shape[1]
        return macs  # per `period` input frames == per output frame


def _rand_w(rng, out_dim, k):
    return (rng.standard_normal((out_dim, k)) / np.sqrt(k)).astype(np.float32)


def _bn(rng, d):
    return (rng.uniform(0.5, 1.5, d).astype(np.float32),
            (0.1 * rng.standard_normal(d)).astype(np.float32))


def make_tdnnf(dim, bottleneck, strides, prefinal_small, num_pdfs, input_dim=40,
               ivector_dim=0, bypass_scale=0.66, seed=3, output_scale=1.0,
               acoustic_scale=1.0, with_priors=True, name="tdnnf"):
    """The xconfig of run_tdnn_1d.sh:219-249 / run_tdnn_1h.sh:163-190 expanded to fused
    layers.  `strides` lists the time-stride of each tdnnf-layer (1,1,1,0,3,3,...)."""
    rng = np.random.default_rng(seed)
    L = []
    spliced = 3 * input_dim + ivector_dim
    # fixed-affine-layer name=lda input=Append(-1,0,1,ReplaceIndex(ivector,t,0))
    L.append(Layer("lda", input_dim, spliced, [-1, 0, 1], -1, _rand_w(rng, spliced, spliced),
                   bias=(0.1 * rng.standard_normal(spliced)).astype(np.float32),
                   ivector_dim=ivector_dim))
    # relu-batchnorm-dropout-layer name=tdnn1
    s, o = _bn(rng, dim)
    L.append(Layer("tdnn1", spliced, dim, [0], 0, _rand_w(rng, dim, spliced),
                   bias=(0.1 * rng.standard_normal(dim)).astype(np.float32), relu=True,
                   bn_scale=s, bn_offset=o))
    prev = 1
    for j, st in enumerate(strides):
        offs1 = [-st, 0] if st != 0 else [0]
        offs2 = [0, st] if st != 0 else [0]
        L.append(Layer("tdnnf%d.linear" % (j + 2), dim, bottleneck, offs1, prev,
                       _rand_w(rng, bottleneck, dim * len(offs1))))
        s, o = _bn(rng, dim)
        L.append(Layer("tdnnf%d.affine" % (j + 2), bottleneck, dim, offs2, len(L) - 1,
                       _rand_w(rng, dim, bottleneck * len(offs2)),
                       bias=(0.1 * rng.standard_normal(dim)).astype(np.float32), relu=True,
                       bn_scale=s, bn_offset=o, bypass_layer=prev, bypass_scale=bypass_scale))
        prev = len(L) - 1
    # linear-component name=prefinal-l
    L.append(Layer("prefinal-l", dim, prefinal_small, [0], prev, _rand_w(rng, prefinal_small, dim)))
    # prefinal-layer name=prefinal-chain: affine, relu, batchnorm1, linear, batchnorm2
    s, o = _bn(rng, dim)
    L.append(Layer("prefinal-chain.affine", prefinal_small, dim, [0], len(L) - 1,
                   _rand_w(rng, dim, prefinal_small),
                   bias=(0.1 * rng.standard_normal(dim)).astype(np.float32), relu=True,
                   bn_scale=s, bn_offset=o))
    s, o = _bn(rng, prefinal_small)
    L.append(Layer("prefinal-chain.linear", dim, prefinal_small, [0], len(L) - 1,
                   _rand_w(rng, prefinal_small, dim), bn_scale=s, bn_offset=o))
    # output-layer name=output include-log-softmax=false
    pri = None
    if with_priors:
        p = rng.dirichlet(np.full(num_pdfs, 5.0))
        pri = (-np.log(p)).astype(np.float32)   # post_offset = -log_priors
    L.append(Layer("output", prefinal_small, num_pdfs, [0], len(L) - 1,
                   (output_scale * _rand_w(rng, num_pdfs, prefinal_small)).astype(np.float32),
                   bias=(0.1 * rng.standard_normal(num_pdfs)).astype(np.float32),
                   post_offset=pri, post_scale=acoustic_scale))
    return Model(L, input_dim, ivector_dim, 3, num_pdfs, name)


def tdnnf_mini_librispeech(num_pdfs=2328, **kw):
    """run_tdnn_1h.sh:163-190: dim 768, bottleneck 96, 12 tdnnf layers, prefinal 192, bypass-scale 0.66 (:157).
    Pinned against the reference's own xconfig generator: tests/test_xconfig_golden.py."""
    return make_tdnnf(768, 96, [1, 1, 1, 0] + [3] * 8, 192, num_pdfs, name="tdnn1h", **kw)


def tdnnf_librispeech(num_pdfs=6000, **kw):
    """run_tdnn_1d.sh:219-249: dim 1536, bottleneck 160, 16 tdnnf layers, prefinal 256, bypass-scale 0.75 (:212).
    Pinned against the reference's own xconfig generator: tests/test_xconfig_golden.py."""
    kw.setdefault("bypass_scale", 0.75)
    return make_tdnnf(1536, 160, [1, 1, 1, 0] + [3] * 12, 256, num_pdfs, name="tdnn1d", **kw)


def tdnnf_tiny(num_pdfs=64, dim=48, bottleneck=16, input_dim=40, ivector_dim=0, seed=5, **kw):
    """Appendix-E sized toy (strides 1,0,3,3) for fast CPU tests."""
    return make_tdnnf(dim, bottleneck, [1, 0, 3, 3], 24, num_pdfs, input_dim=input_dim,
                      ivector_dim=ivector_dim, seed=seed, name="tiny", **kw)
