"""oracle/orc_blas.py -- TEST / BASELINE INFRASTRUCTURE ONLY (never shipped; only tests/ and bench.py's cpu_baseline
leg import it).

The nnet3 forward the way the REFERENCE runs it on a CPU: `DecodableNnetSimple` evaluates the utterance chunk by chunk
(nnet3/nnet-am-decodable-simple.cc:93-167: `frames_per_chunk` = 50 input frames rounded up to a multiple of the
frame-subsampling factor (:278-310), left / right context frames around every chunk, first / last frame repeated at the
utterance edges :147-160, the context rows RECOMPUTED for every chunk), and inside a chunk every component's
Propagate is one BLAS sgemm over all the rows of the chunk that are needed (matrix/kaldi-matrix.cc:182 AddMatMat ->
cblas_sgemm, matrix/cblas-wrappers.h:233; TdnnComponent: one sgemm per time offset, nnet-tdnn-component.cc:201-210;
AffineComponent nnet-simple-component.cc:1234-1243), each node evaluated only at the time indexes its consumers
request (the compiled computation, nnet-compile.cc).  Here the sgemm is numpy's float32 matmul (OpenBLAS in this
image; the caller pins it to one BLAS thread per worker: nnet3-latgen-faster is single-threaded per job).

Same fused-layer model as oracle/orc_nnet.cc (the scalar oracle used for parity), same results to fp32 rounding
(tests/test_oracle_nnet.py::test_blas_forward_equals_scalar_oracle); this one exists so that the CPU baseline of
bench.py times the reference's arithmetic path (sgemm), not a scalar loop.
"""
import numpy as np


def _chunk_plan(model, t_out):
    """Times every layer must be evaluated at for the output times t_out (sorted int arrays; index -1 = input)."""
    n = len(model.layers)
    if any(l.slice_layers is not None for l in model.layers):
        raise ValueError("the sgemm baseline evaluates single-producer layers only (use orc.nnet_forward)")
    req = {n - 1: np.asarray(t_out, np.int64)}
    for i in range(n - 1, -1, -1):
        l = model.layers[i]
        if i not in req:
            raise ValueError("layer %d has no consumer" % i)
        t = req[i]
        need = np.unique(np.concatenate([t + o for o in l.offsets]))
        req[l.input_layer] = need if l.input_layer not in req else np.union1d(req[l.input_layer], need)
        if l.bypass_layer != -2:
            req[l.bypass_layer] = t if l.bypass_layer not in req else np.union1d(req[l.bypass_layer], t)
    return req


def _forward_times(model, feats, ivector, t_out, weights):
    T = feats.shape[0]
    req = _chunk_plan(model, t_out)
    acts = {-1: feats[np.clip(req[-1], 0, T - 1)]}
    for i, l in enumerate(model.layers):
        t = req[i]
        src, src_t = acts[l.input_layer], req[l.input_layer]
        Ws, Wiv = weights[i]
        y = None
        for j, o in enumerate(l.offsets):
            x = src[np.searchsorted(src_t, t + o)]
            part = x @ Ws[j]                                  # sgemm: [rows x in_dim] . [in_dim x out_dim]
            y = part if y is None else y + part
        if l.bias is not None:
            y = y + l.bias
        if Wiv is not None:
            y = y + (ivector @ Wiv)[None, :]
        if l.relu:
            np.maximum(y, 0.0, out=y)
        if l.bn_scale is not None:
            y = y * l.bn_scale + l.bn_offset
        if l.bypass_layer != -2:
            y = y + np.float32(l.bypass_scale) * acts[l.bypass_layer][np.searchsorted(req[l.bypass_layer], t)]
        if l.log_softmax:
            m = y.max(axis=1, keepdims=True)
            y = y - (m + np.log(np.exp(y - m).sum(axis=1, keepdims=True)))
        if l.post_offset is not None:
            y = y + l.post_offset
        if l.post_scale != 1.0:
            y = y * np.float32(l.post_scale)
        acts[i] = y.astype(np.float32, copy=False)
    return acts[len(model.layers) - 1]


def prepare(model):
    """Per-layer transposed weight slices (once per model): [(per-offset [in_dim x out_dim] float32, ivector part or None)]."""
    out = []
    for l in model.layers:
        W = np.asarray(l.W, np.float32)
        k = len(l.offsets) * l.in_dim
        Ws = [np.ascontiguousarray(W[:, j * l.in_dim:(j + 1) * l.in_dim].T) for j in range(len(l.offsets))]
        Wiv = np.ascontiguousarray(W[:, k:].T) if l.ivector_dim else None
        out.append((Ws, Wiv))
    return out


def nnet_forward_blas(model, feats, ivector=None, frames_per_chunk=50, weights=None):
    """[ceil(T / subsampling) x P] log-likelihoods, chunk by chunk like DecodableNnetSimple (frames_per_chunk <= 0:
    the whole utterance as one chunk)."""
    feats = np.ascontiguousarray(feats, np.float32)
    T, sub = feats.shape[0], model.subsampling
    n_out = (T + sub - 1) // sub
    if weights is None:
        weights = prepare(model)
    iv = None if ivector is None else np.asarray(ivector, np.float32)
    if frames_per_chunk <= 0:
        C = n_out
    else:
        C = (frames_per_chunk + sub - 1) // sub              # CheckAndFixConfigs: rounded up to a multiple of `sub`
    out = np.empty((n_out, model.layers[-1].out_dim), np.float32)
    for start in range(0, n_out, C):
        num = min(C, n_out - start)
        t_out = (start + np.arange(num)) * sub
        out[start:start + num] = _forward_times(model, feats, iv, t_out, weights)
    return out
