"""Independent float64 numpy evaluation of the fused-layer TDNN(-F) stack (dense over a
padded time range; edge frames clamped at the INPUT only, as
nnet3/nnet-am-decodable-simple.cc:147-160 does)."""
import numpy as np


def forward_f64(model, feats, ivector=None):
    T = feats.shape[0]
    left, right = model.context()
    sub = model.subsampling
    n_out = (T + sub - 1) // sub
    t_lo, t_hi = -left, (n_out - 1) * sub + right        # dense time range of interest
    pad = 64
    times = np.arange(t_lo - pad, t_hi + pad + 1)
    x = feats[np.clip(times, 0, T - 1)].astype(np.float64)
    acts = {-1: x}
    for i, l in enumerate(model.layers):
        n = x.shape[0]
        W = l.W.astype(np.float64)
        y = np.zeros((n, l.out_dim))
        for prod, off, col, width in l.slices():        # (one producer for all slices, or Append over different ones)
            Wj = W[:, col:col + width]
            shifted = np.roll(acts[prod], -off, axis=0)  # row r holds src[t_r + off]
            y += shifted @ Wj.T
        if l.ivector_dim:
            y += (W[:, W.shape[1] - l.ivector_dim:] @ ivector.astype(np.float64))[None, :]
        if l.bias is not None:
            y += l.bias.astype(np.float64)
        if l.relu:
            y = np.maximum(y, 0.0)
        if l.bn_scale is not None:
            y = y * l.bn_scale.astype(np.float64) + l.bn_offset.astype(np.float64)
        if l.bypass_layer != -2:
            y = y + l.bypass_scale * acts[l.bypass_layer]
        if getattr(l, 'log_softmax', False):
            m = y.max(axis=1, keepdims=True)
            y = y - (m + np.log(np.exp(y - m).sum(axis=1, keepdims=True)))
        if l.post_offset is not None:
            y = y + l.post_offset.astype(np.float64)
        y = y * l.post_scale
        acts[i] = y
    out_t = np.arange(n_out) * sub
    idx = out_t - times[0]
    return acts[len(model.layers) - 1][idx]
