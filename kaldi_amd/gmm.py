"""GMM acoustic models (BASELINE configs[0]: egs/yesno monophone GMM, gmm-latgen-faster): the model file of
gmmbin/gmm-latgen-faster.cc:90-100 = TransitionModel + AmDiagGmm (gmm/am-diag-gmm.cc:147-176, gmm/diag-gmm.cc:705-756),
and DecodableAmDiagGmmScaled's log-likelihood matrix on the device (kaldi_amd/csrc/gmm.hip)."""
import ctypes as C
import struct

import numpy as np

from . import abi, mdl
from ._lib import KamdError, check, lib


class AmDiagGmm:
    def __init__(self, weights, means_invvars, inv_vars):
        """lists, one entry per pdf: weights [M], means_invvars / inv_vars [M x dim]"""
        self.num_pdfs = len(weights)
        self.dim = means_invvars[0].shape[1]
        self.mix_off = np.concatenate([[0], np.cumsum([w.size for w in weights])]).astype(np.int32)
        self.weights = np.ascontiguousarray(np.concatenate(weights), np.float32)
        self.means_invvars = np.ascontiguousarray(np.concatenate(means_invvars), np.float32)
        self.inv_vars = np.ascontiguousarray(np.concatenate(inv_vars), np.float32)
        if self.means_invvars.shape != self.inv_vars.shape or self.means_invvars.shape[0] != self.weights.size:
            raise KamdError("AmDiagGmm: inconsistent sizes")
        # DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-150): recomputed on reading, as the reference does
        D = self.dim
        gc = np.log(self.weights.astype(np.float32)) + np.float32(-0.5 * 1.8378770664093453 * D)
        gc = gc + (np.float32(0.5) * np.log(self.inv_vars) - np.float32(0.5) * self.means_invvars ** 2 / self.inv_vars).sum(1, dtype=np.float32)
        self.gconsts = np.where(np.isposinf(gc), -gc, gc).astype(np.float32)

    def pdf(self, p):
        a, b = self.mix_off[p], self.mix_off[p + 1]
        return self.weights[a:b], self.means_invvars[a:b], self.inv_vars[a:b]


def read_gmm_mdl(path):
    """-> (AmDiagGmm, id2pdf, tid_phone, tid2phone) from a binary GMM final.mdl"""
    s = mdl._Stream(open(path, "rb").read())
    if s.take(2) != b"\0B":
        raise KamdError("binary Kaldi file expected")
    id2pdf, tid_phone, _ = mdl.read_transition_model(s)
    tid2phone = mdl.read_transition_model.tid2phone
    s.expect("<DIMENSION>")
    dim = s.i32()
    s.expect("<NUMPDFS>")
    n = s.i32()
    W, MIV, IV = [], [], []
    for _ in range(n):
        t = s.token()
        if t not in ("<DiagGMM>", "<DiagGMMBegin>"):
            raise KamdError("Expected <DiagGMM>, got " + t)
        t = s.token()
        if t == "<GCONSTS>":
            s.vector()
            t = s.token()
        if t != "<WEIGHTS>":
            raise KamdError("DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " + t)
        W.append(s.vector())
        s.expect("<MEANS_INVVARS>")
        MIV.append(s.matrix())
        s.expect("<INV_VARS>")
        IV.append(s.matrix())
        t = s.token()
        if t not in ("</DiagGMM>", "<DiagGMMEnd>"):
            raise KamdError("Expected </DiagGMM>, got " + t)
        if MIV[-1].shape[1] != dim:
            raise KamdError("AmDiagGmm: pdf dimension mismatch")
    return AmDiagGmm(W, MIV, IV), id2pdf, tid_phone, tid2phone


def write_am_diag_gmm(f, am):
    """AmDiagGmm::Write, binary (for the tests: no GMM model exists offline)"""
    def tok(t):
        f.write(t.encode() + b" ")

    def i32(v):
        f.write(b"\x04" + struct.pack("<i", v))

    def vec(a):
        tok("FV"); i32(a.size); f.write(np.ascontiguousarray(a, np.float32).tobytes())

    def mat(a):
        tok("FM"); i32(a.shape[0]); i32(a.shape[1]); f.write(np.ascontiguousarray(a, np.float32).tobytes())
    tok("<DIMENSION>"); i32(am.dim); tok("<NUMPDFS>"); i32(am.num_pdfs)
    for p in range(am.num_pdfs):
        w, miv, iv = am.pdf(p)
        a, b = am.mix_off[p], am.mix_off[p + 1]
        tok("<DiagGMM>"); tok("<GCONSTS>"); vec(am.gconsts[a:b]); tok("<WEIGHTS>"); vec(w)
        tok("<MEANS_INVVARS>"); mat(miv); tok("<INV_VARS>"); mat(iv); tok("</DiagGMM>")


class DecodableAmDiagGmmScaled:
    """the device handle: loglikes(feats, scale) -> [T x num_pdfs]"""

    def __init__(self, am):
        self.am = am
        self._h = lib().kamd_am_gmm_create(am.num_pdfs, am.dim, abi.iptr(am.mix_off), abi.fptr(am.gconsts), abi.fptr(am.means_invvars),
                                           abi.fptr(am.inv_vars))
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_am_gmm_destroy(self._h)
            self._h = None

    def loglikes(self, feats, scale=1.0):
        f = np.ascontiguousarray(feats, np.float32)
        out = np.zeros((f.shape[0], self.am.num_pdfs), np.float32)
        check(lib().kamd_am_gmm_loglikes(self._h, abi.fptr(f), f.shape[0], f.shape[1], scale, abi.fptr(out)))
        return out
