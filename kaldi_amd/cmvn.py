"""compute-cmvn-stats / apply-cmvn (transform/cmvn.cc:30-118, featbin/apply-cmvn.cc) on the device.
Statistics are [2 x (dim+1)] float64 matrices as Kaldi stores them: row 0 sums and, in the last
column, the count; row 1 sums of squares."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import KamdError, check, lib


class _Dev:
    def __init__(self, host):
        self.n = host.nbytes
        self.p = lib().kamd_malloc(max(self.n, 4))
        if not self.p:
            raise KamdError(lib().kamd_last_error().decode())
        check(lib().kamd_memcpy_h2d(self.p, host.ctypes.data_as(C.c_void_p), self.n))

    def download(self, out):
        check(lib().kamd_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.p, self.n))

    def __del__(self):
        if getattr(self, "p", None):
            lib().kamd_free(self.p)
            self.p = None


def _batch(mats):
    mats = [np.ascontiguousarray(m, np.float32) for m in mats]
    dim = mats[0].shape[1]
    if any(m.ndim != 2 or m.shape[1] != dim for m in mats):
        raise KamdError("cmvn: feature matrices must share one dimension")
    off = np.zeros(len(mats) + 1, np.int64)
    off[1:] = np.cumsum([m.shape[0] for m in mats])
    return mats, np.ascontiguousarray(np.concatenate(mats) if off[-1] else np.zeros((0, dim), np.float32)), off, dim


def acc_stats(mats, stats=None, weights=None):
    """AccCmvnStats for every matrix; stats (optional): [n, 2, dim+1] running statistics to add to; weights (optional):
    one float vector per matrix, a weight per frame (compute-cmvn-stats --weights)."""
    mats, flat, off, dim = _batch(mats)
    st = np.zeros((len(mats), 2, dim + 1), np.float64) if stats is None else np.array(stats, np.float64).reshape(len(mats), 2, dim + 1)
    if off[-1] == 0:
        return st
    d = _Dev(flat)
    if weights is None:
        check(lib().kamd_cmvn_acc_stats_device(d.p, abi.iptr(off, C.c_int64), dim, dim, len(mats), st.ctypes.data_as(C.POINTER(C.c_double)), None))
        return st
    weights = [np.ascontiguousarray(w, np.float32).reshape(-1) for w in weights]
    if len(weights) != len(mats) or any(w.size != m.shape[0] for w, m in zip(weights, mats)):
        raise KamdError("cmvn: one weight per frame is needed (weights->Dim() == num_frames)")
    dw = _Dev(np.ascontiguousarray(np.concatenate(weights)))
    check(lib().kamd_cmvn_acc_stats_weighted_device(d.p, abi.iptr(off, C.c_int64), dim, dim, len(mats), dw.p,
                                                    st.ctypes.data_as(C.POINTER(C.c_double)), None))
    return st


def fake_stats_for_some_dims(stats, dims):
    """FakeStatsForSomeDims (transform/cmvn.cc:171-182): statistics [..., 2, dim+1] under which the listed
    dimensions stay as they are (mean 0, variance 1) -- apply-cmvn --skip-dims"""
    st = np.array(stats, np.float64)
    dim = st.shape[-1] - 1
    for d in dims:
        if not 0 <= int(d) < dim:
            raise KamdError("skip-dims: dimension %d out of range (feature dim %d)" % (int(d), dim))
        st[..., 0, int(d)] = 0.0
        st[..., 1, int(d)] = st[..., 0, dim]
    return st


def apply(mats, stats, norm_means=True, norm_vars=False, reverse=False, skip_dims=()):
    """ApplyCmvn (reverse: ApplyCmvnReverse): matrix i normalised with stats[i] ([2, dim+1]); returns new matrices."""
    mats, flat, off, dim = _batch(mats)
    st = np.ascontiguousarray(stats, np.float64).reshape(len(mats), 2, dim + 1)
    if len(skip_dims):
        st = np.ascontiguousarray(fake_stats_for_some_dims(st, skip_dims))
    if off[-1] == 0:
        return [m.copy() for m in mats]
    d = _Dev(flat)
    fn = lib().kamd_cmvn_apply_reverse_device if reverse else lib().kamd_cmvn_apply_device
    check(fn(d.p, abi.iptr(off, C.c_int64), dim, dim, len(mats), st.ctypes.data_as(C.POINTER(C.c_double)), int(norm_means), int(norm_vars), None))
    out = np.zeros_like(flat)
    d.download(out)
    return [out[off[i]:off[i + 1]].copy() for i in range(len(mats))]


def add_deltas(mats, order=2, window=2):
    """add-deltas (featbin/add-deltas.cc, ComputeDeltas): [T x dim] -> [T x (order+1)*dim] per matrix"""
    mats, flat, off, dim = _batch(mats)
    if off[-1] == 0:
        return [np.zeros((0, (order + 1) * dim), np.float32) for _ in mats]
    d_in = _Dev(flat)
    out = np.zeros((flat.shape[0], (order + 1) * dim), np.float32)
    d_out = _Dev(out)
    check(lib().kamd_feat_add_deltas_device(d_in.p, dim, d_out.p, (order + 1) * dim, abi.iptr(off, C.c_int64), len(mats), dim, order, window, None))
    d_out.download(out)
    return [out[off[i]:off[i + 1]].copy() for i in range(len(mats))]


def splice_transform(mats, left=0, right=0, transforms=None, utt_transform=None):
    """splice-feats (left / right context, clamped) and / or transform-feats (linear or affine; `transforms`: one matrix for
    all, or a list with utt_transform[i] = index of matrix i's transform) -> new matrices"""
    mats, flat, off, dim = _batch(mats)
    sd = dim * (left + 1 + right)
    xf = None
    if transforms is not None:
        tl = [transforms] if np.asarray(transforms).ndim == 2 else list(transforms)
        xf = np.ascontiguousarray(np.stack([np.asarray(t, np.float32) for t in tl]))
        ux = np.zeros(len(mats), np.int32) if utt_transform is None else np.ascontiguousarray(utt_transform, np.int32)
    n_out = xf.shape[1] if xf is not None else sd
    if off[-1] == 0:
        return [np.zeros((0, n_out), np.float32) for _ in mats]
    d_in = _Dev(flat)
    out = np.zeros((flat.shape[0], n_out), np.float32)
    d_out = _Dev(out)
    check(lib().kamd_feat_splice_transform_device(d_in.p, dim, d_out.p, n_out, abi.iptr(off, C.c_int64), len(mats), dim, left, right,
                                                  abi.fptr(xf) if xf is not None else None, xf.shape[0] if xf is not None else 0,
                                                  abi.iptr(ux) if xf is not None else None, xf.shape[1] if xf is not None else 0,
                                                  xf.shape[2] if xf is not None else 0, None))
    d_out.download(out)
    return [out[off[i]:off[i + 1]].copy() for i in range(len(mats))]
