"""Online i-vector extraction (host mirror of online2/online-ivector-feature.h's
OnlineIvectorExtractionInfo + OnlineIvectorFeature as ivector-extract-online2 runs it).

  info = IvectorExtractionInfo.from_config("conf/ivector_extractor.conf")   # final.ie, final.dubm, final.mat, ...
  ie = IvectorExtractor(info)                    # uploads the model; the HIP library does the work
  ivectors = ie.extract_online(feats)            # [ceil(T / period) x dim], what --online-ivectors takes

The file readers follow IvectorExtractor::Write (ivector/ivector-extractor.cc:802-826), DiagGmm::Write
(gmm/diag-gmm.cc:690-726), PackedMatrix::Write (matrix/packed-matrix.cc:236-271) and the Matrix /
Vector writers; write_* are their inverse, used by the tests (no extractor is available offline)."""
import ctypes as C
import math
import os
import struct

import numpy as np

from . import abi, table
from ._lib import KamdError, check, lib


class IvectorExtractionInfo:
    """The fields of OnlineIvectorExtractionInfo (online2/online-ivector-feature.h:140-185)."""

    def __init__(self, lda, global_cmvn_stats, ubm_weights, ubm_means_invvars, ubm_inv_vars, M, sigma_inv, prior_offset,
                 splice_left=3, splice_right=3, cmn_window=600, speaker_frames=600, global_frames=200, normalize_mean=True,
                 normalize_variance=False, ivector_period=10, num_gselect=5, min_post=0.025, posterior_scale=0.1,
                 max_count=0.0, num_cg_iters=15):
        self.lda = np.ascontiguousarray(lda, np.float32)
        self.global_cmvn_stats = np.ascontiguousarray(global_cmvn_stats, np.float64)
        self.ubm_weights = np.ascontiguousarray(ubm_weights, np.float32)
        self.ubm_means_invvars = np.ascontiguousarray(ubm_means_invvars, np.float32)
        self.ubm_inv_vars = np.ascontiguousarray(ubm_inv_vars, np.float32)
        self.M = np.ascontiguousarray(M, np.float64)                     # [G, D, I]
        self.sigma_inv = np.ascontiguousarray(sigma_inv, np.float64)     # [G, D(D+1)/2] packed lower triangle
        self.prior_offset = float(prior_offset)
        self.splice_left, self.splice_right = int(splice_left), int(splice_right)
        self.cmn_window, self.speaker_frames, self.global_frames = int(cmn_window), int(speaker_frames), int(global_frames)
        self.normalize_mean, self.normalize_variance = bool(normalize_mean), bool(normalize_variance)
        self.ivector_period, self.num_gselect, self.num_cg_iters = int(ivector_period), int(num_gselect), int(num_cg_iters)
        self.min_post, self.posterior_scale, self.max_count = float(min_post), float(posterior_scale), float(max_count)
        self.check()
        self.ubm_gconsts = self._gconsts()

    # DiagGmm::ComputeGconsts (gmm/diag-gmm.cc:114-150), float arithmetic as there
    def _gconsts(self):
        D = self.ubm_inv_vars.shape[1]
        out = np.zeros(self.ubm_weights.size, np.float32)
        offset = np.float32(-0.5 * 1.8378770664093453 * D)
        for g in range(out.size):
            # (logs through libm in double, rounded to float: bit-identical with csrc/ivector_io.cc)
            gc = np.float32(math.log(float(self.ubm_weights[g])) if self.ubm_weights[g] > 0 else -np.inf) + offset
            for k in range(D):
                iv, mi = self.ubm_inv_vars[g, k], self.ubm_means_invvars[g, k]
                gc = np.float32(gc + np.float32(np.float32(0.5) * np.float32(math.log(float(iv)))) - np.float32(np.float32(0.5) * mi * mi / iv))
            if np.isinf(gc) and gc > 0:
                gc = -gc
            out[g] = gc
        return out

    def check(self):
        """OnlineIvectorExtractionInfo::Check (online2/online-ivector-feature.cc:76-93) + what the device path needs"""
        if self.global_cmvn_stats.shape[0] != 2:
            raise KamdError("global CMVN stats must have two rows")
        self.feat_dim = self.global_cmvn_stats.shape[1] - 1
        sd = self.feat_dim * (self.splice_left + 1 + self.splice_right)
        if self.lda.shape[1] not in (sd, sd + 1):
            raise KamdError("LDA matrix has %d columns, spliced features %d" % (self.lda.shape[1], sd))
        D = self.lda.shape[0]
        G = self.ubm_weights.size
        if self.ubm_means_invvars.shape != (G, D) or self.ubm_inv_vars.shape != (G, D):
            raise KamdError("diagonal UBM does not match the LDA output dimension")
        if self.M.ndim != 3 or self.M.shape[:2] != (G, D) or self.sigma_inv.shape != (G, D * (D + 1) // 2):
            raise KamdError("i-vector extractor does not match the UBM")
        if not (self.ivector_period > 0 and self.num_gselect > 0 and self.min_post < 0.5 and 0 < self.posterior_scale <= 1.0):
            raise KamdError("bad i-vector extraction options")
        if not (self.speaker_frames <= self.cmn_window and self.global_frames <= self.speaker_frames):
            raise KamdError("OnlineCmvnOptions::Check failed")
        self.ivector_dim = self.M.shape[2]

    def state_size(self):
        """doubles in an adaptation state: 2 x (feat_dim+1) speaker CMVN stats, packed quadratic term, linear term, count"""
        return 2 * (self.feat_dim + 1) + self.ivector_dim * (self.ivector_dim + 1) // 2 + self.ivector_dim + 1

    def desc(self):
        d = abi.IvectorDesc()
        d.feat_dim, d.splice_left, d.splice_right = self.feat_dim, self.splice_left, self.splice_right
        d.lda_rows, d.lda_cols = self.lda.shape
        d.lda = abi.fptr(self.lda)
        d.global_cmvn_stats = self.global_cmvn_stats.ctypes.data_as(C.POINTER(C.c_double))
        d.cmn_window, d.speaker_frames, d.global_frames = self.cmn_window, self.speaker_frames, self.global_frames
        d.normalize_mean, d.normalize_variance = int(self.normalize_mean), int(self.normalize_variance)
        d.num_gauss = self.ubm_weights.size
        d.ubm_gconsts, d.ubm_means_invvars, d.ubm_inv_vars = abi.fptr(self.ubm_gconsts), abi.fptr(self.ubm_means_invvars), abi.fptr(self.ubm_inv_vars)
        d.ivector_dim = self.ivector_dim
        d.M = self.M.ctypes.data_as(C.POINTER(C.c_double))
        d.sigma_inv = self.sigma_inv.ctypes.data_as(C.POINTER(C.c_double))
        d.prior_offset = self.prior_offset
        d.ivector_period, d.num_gselect, d.num_cg_iters = self.ivector_period, self.num_gselect, self.num_cg_iters
        d.min_post, d.posterior_scale, d.max_count = self.min_post, self.posterior_scale, self.max_count
        return d

    # ---- files ------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config_rxfilename):
        """--ivector-extraction-config of the online2 binaries / --config of ivector-extract-online2:
        OnlineIvectorExtractionConfig::Register (online2/online-ivector-feature.h:90-137)."""
        po = table.ParseOptions("ivector extraction config")
        for name in ("lda-matrix", "global-cmvn-stats", "cmvn-config", "splice-config", "diag-ubm", "ivector-extractor"):
            po.register(name, str, "")
        po.register("ivector-period", int, 10); po.register("num-gselect", int, 5); po.register("min-post", float, 0.025)
        po.register("posterior-scale", float, 0.1); po.register("max-count", float, 0.0)
        po.register("use-most-recent-ivector", bool, True); po.register("greedy-ivector-extractor", bool, False)
        po.register("max-remembered-frames", float, 1000.0)
        po.read_config_file(config_rxfilename)
        for name in ("lda-matrix", "global-cmvn-stats", "cmvn-config", "splice-config", "diag-ubm", "ivector-extractor"):
            if not po[name]:
                raise KamdError("--%s option must be set (note: this may be needed in the file supplied to "
                                "--ivector-extractor-config)" % name)
        cm = table.ParseOptions("online cmvn config")
        cm.register("cmn-window", int, 600); cm.register("global-frames", int, 200); cm.register("speaker-frames", int, 600)
        cm.register("norm-vars", bool, False); cm.register("norm-means", bool, True); cm.register("skip-dims", str, "")
        cm.read_config_file(po["cmvn-config"])
        if cm["skip-dims"]:
            raise KamdError("--skip-dims is not supported")
        sp = table.ParseOptions("splice config")
        sp.register("left-context", int, 4); sp.register("right-context", int, 4)
        sp.read_config_file(po["splice-config"])
        w, miv, iv = read_diag_gmm(po["diag-ubm"])
        M, sinv, off = read_ivector_extractor(po["ivector-extractor"])
        return cls(read_kaldi_matrix(po["lda-matrix"]), read_kaldi_matrix(po["global-cmvn-stats"], np.float64), w, miv, iv, M, sinv,
                   off, sp["left-context"], sp["right-context"], cm["cmn-window"], cm["speaker-frames"], cm["global-frames"],
                   cm["norm-means"], cm["norm-vars"], po["ivector-period"], po["num-gselect"], po["min-post"],
                   po["posterior-scale"], po["max-count"])


def read_config_native(config_rxfilename):
    """The same files through the library's own reader (kamd_ivector_info_read, csrc/ivector_io.cc: what a C / C++ host
    uses) -> IvectorExtractionInfo; every field bit-identical with from_config (tests/test_ivector_io.py)."""
    L = lib()
    h = L.kamd_ivector_info_read(str(config_rxfilename).encode())
    if not h:
        raise KamdError(L.kamd_last_error().decode())
    try:
        d = L.kamd_ivector_info_desc(h).contents
        G, D, I, sdim = d.num_gauss, d.lda_rows, d.ivector_dim, d.feat_dim + 1
        arr = lambda p, *shape: np.ctypeslib.as_array(p, shape=shape).copy()
        info = IvectorExtractionInfo(arr(d.lda, D, d.lda_cols), arr(d.global_cmvn_stats, 2, sdim), np.ones(G, np.float32),
                                     arr(d.ubm_means_invvars, G, D), arr(d.ubm_inv_vars, G, D), arr(d.M, G, D, I),
                                     arr(d.sigma_inv, G, D * (D + 1) // 2), d.prior_offset, d.splice_left, d.splice_right, d.cmn_window,
                                     d.speaker_frames, d.global_frames, bool(d.normalize_mean), bool(d.normalize_variance), d.ivector_period,
                                     d.num_gselect, d.min_post, d.posterior_scale, d.max_count, d.num_cg_iters)
        info.ubm_gconsts = arr(d.ubm_gconsts, G)          # (the weights themselves are not part of the descriptor)
        return info
    finally:
        L.kamd_ivector_info_destroy(h)


# ---- Kaldi object files ---------------------------------------------------------------------
class _In:
    def __init__(self, rxfilename):
        with table.Input(rxfilename) as (path, off):
            with open(path, "rb") as f:
                f.seek(off)
                self.b = f.read()
        if self.b[:2] != b"\0B":
            raise KamdError("%s: only binary-mode Kaldi objects are read" % rxfilename)
        self.p = 2

    def token(self):
        e = self.b.index(b" ", self.p)
        t = self.b[self.p:e].decode()
        self.p = e + 1
        return t

    def expect(self, *toks):
        t = self.token()
        if t not in toks:
            raise KamdError("expected %s, got %s" % (" or ".join(toks), t))
        return t

    def i32(self):
        if self.b[self.p] != 4:
            raise KamdError("int32 expected")
        v = struct.unpack_from("<i", self.b, self.p + 1)[0]
        self.p += 5
        return v

    def real(self):
        n = self.b[self.p]
        v = struct.unpack_from("<f" if n == 4 else "<d", self.b, self.p + 1)[0]
        self.p += 1 + n
        return v

    def _data(self, n, tok):
        w = 4 if tok[0] == "F" else 8
        a = np.frombuffer(self.b, "<f4" if w == 4 else "<f8", n, self.p).copy()
        self.p += w * n
        return a

    def vector(self):
        t = self.expect("FV", "DV")
        return self._data(self.i32(), t)

    def matrix(self):
        t = self.expect("FM", "DM")
        r, c = self.i32(), self.i32()
        return self._data(r * c, t).reshape(r, c)

    def packed(self):
        t = self.expect("FP", "DP")
        n = self.i32()
        return self._data(n * (n + 1) // 2, t)


def read_kaldi_matrix(rxfilename, dtype=np.float32):
    return np.ascontiguousarray(_In(rxfilename).matrix(), dtype)


def read_diag_gmm(rxfilename):
    """-> (weights, means_invvars, inv_vars); the stored gconsts are recomputed, as DiagGmm::Read does"""
    s = _In(rxfilename)
    s.expect("<DiagGMM>", "<DiagGMMBegin>")
    t = s.token()
    if t == "<GCONSTS>":
        s.vector()
        t = s.token()
    if t != "<WEIGHTS>":
        raise KamdError("DiagGmm::Read, expected <WEIGHTS> or <GCONSTS>, got " + t)
    w = s.vector()
    s.expect("<MEANS_INVVARS>")
    miv = s.matrix()
    s.expect("<INV_VARS>")
    iv = s.matrix()
    s.expect("</DiagGMM>", "<DiagGMMEnd>")
    return w.astype(np.float32), miv.astype(np.float32), iv.astype(np.float32)


def read_ivector_extractor(rxfilename):
    """-> (M[G, D, I], Sigma_inv[G, D(D+1)/2], prior_offset).  Extractors with i-vector dependent
    weights (a non-empty <w>) are not what the online recipes train and are rejected, as
    OnlineIvectorEstimationStats::AccStats asserts."""
    s = _In(rxfilename)
    s.expect("<IvectorExtractor>")
    s.expect("<w>")
    w = s.matrix()
    if w.size:
        raise KamdError("i-vector dependent weights are not supported by the online extractor")
    s.expect("<w_vec>")
    s.vector()
    s.expect("<M>")
    G = s.i32()
    M = np.stack([s.matrix() for _ in range(G)]).astype(np.float64)
    s.expect("<SigmaInv>")
    sinv = np.stack([s.packed() for _ in range(G)]).astype(np.float64)
    s.expect("<IvectorOffset>")
    off = s.real()
    s.expect("</IvectorExtractor>")
    return M, sinv, off


def _tok(t):
    return t.encode() + b" "


def _i32(v):
    return b"\x04" + struct.pack("<i", v)


def _mat(a):
    a = np.asarray(a)
    d = a.dtype == np.float64
    return _tok("DM" if d else "FM") + _i32(a.shape[0]) + _i32(a.shape[1] if a.ndim == 2 else 0) + np.ascontiguousarray(a).tobytes()


def _vec(a):
    a = np.asarray(a)
    return _tok("DV" if a.dtype == np.float64 else "FV") + _i32(a.size) + np.ascontiguousarray(a).tobytes()


def write_kaldi_matrix(path, a):
    with open(path, "wb") as f:
        f.write(b"\0B" + _mat(a))


def write_diag_gmm(path, info):
    with open(path, "wb") as f:
        f.write(b"\0B" + _tok("<DiagGMM>") + _tok("<GCONSTS>") + _vec(info.ubm_gconsts) + _tok("<WEIGHTS>") + _vec(info.ubm_weights) +
                _tok("<MEANS_INVVARS>") + _mat(info.ubm_means_invvars) + _tok("<INV_VARS>") + _mat(info.ubm_inv_vars) + _tok("</DiagGMM>"))


def write_ivector_extractor(path, info):
    G = info.M.shape[0]
    D = info.M.shape[1]
    with open(path, "wb") as f:
        f.write(b"\0B" + _tok("<IvectorExtractor>") + _tok("<w>") + _mat(np.zeros((0, 0), np.float64)) + _tok("<w_vec>") +
                _vec(np.log(np.full(G, 1.0 / G))) + _tok("<M>") + _i32(G))
        for g in range(G):
            f.write(_mat(info.M[g]))
        f.write(_tok("<SigmaInv>"))
        for g in range(G):
            f.write(_tok("DP") + _i32(D) + info.sigma_inv[g].tobytes())
        f.write(_tok("<IvectorOffset>") + b"\x08" + struct.pack("<d", info.prior_offset) + _tok("</IvectorExtractor>"))


def write_config_dir(dirname, info):
    """final.ie, final.dubm, final.mat, global_cmvn.stats, online_cmvn.conf, splice.conf and
    ivector_extractor.conf as steps/online/nnet2/prepare_online_decoding.sh lays them out."""
    os.makedirs(dirname, exist_ok=True)
    p = lambda n: os.path.join(str(dirname), n)
    write_ivector_extractor(p("final.ie"), info)
    write_diag_gmm(p("final.dubm"), info)
    write_kaldi_matrix(p("final.mat"), info.lda)
    write_kaldi_matrix(p("global_cmvn.stats"), info.global_cmvn_stats)
    open(p("online_cmvn.conf"), "w").write("# configuration file for apply-cmvn-online\n--cmn-window=%d\n--speaker-frames=%d\n--global-frames=%d\n" %
                                           (info.cmn_window, info.speaker_frames, info.global_frames))
    open(p("splice.conf"), "w").write("--left-context=%d\n--right-context=%d\n" % (info.splice_left, info.splice_right))
    open(p("ivector_extractor.conf"), "w").write(
        "--cmvn-config=%s\n--ivector-period=%d\n--splice-config=%s\n--lda-matrix=%s\n--global-cmvn-stats=%s\n--diag-ubm=%s\n"
        "--ivector-extractor=%s\n--num-gselect=%d\n--min-post=%g\n--posterior-scale=%g\n--max-remembered-frames=1000\n--max-count=%g\n" %
        (p("online_cmvn.conf"), info.ivector_period, p("splice.conf"), p("final.mat"), p("global_cmvn.stats"), p("final.dubm"),
         p("final.ie"), info.num_gselect, info.min_post, info.posterior_scale, info.max_count))
    return p("ivector_extractor.conf")


def make_synthetic(feat_dim=40, lda_dim=40, num_gauss=512, ivector_dim=100, seed=0, feat_mean=None, feat_std=None, **opts):
    """A random but well-conditioned extractor of the recipe's shape (hires MFCC 40 -> splice +-3 ->
    LDA 40 -> 512-Gaussian UBM -> 100-dim i-vectors); feat_mean / feat_std describe the features
    it will see (the global CMVN stats and the UBM are placed accordingly)."""
    rng = np.random.default_rng(seed)
    L = opts.pop("splice_left", 3)
    R = opts.pop("splice_right", 3)
    sd = feat_dim * (L + 1 + R)
    mean = np.zeros(feat_dim) if feat_mean is None else np.asarray(feat_mean, np.float64)
    std = np.ones(feat_dim) if feat_std is None else np.asarray(feat_std, np.float64)
    lda = (rng.standard_normal((lda_dim, sd + 1)) / np.sqrt(sd)).astype(np.float32)
    lda[:, :sd] /= np.tile(std, L + 1 + R).astype(np.float32)
    lda[:, sd] = 0.1 * rng.standard_normal(lda_dim)
    cnt = 1.0e5
    gstats = np.zeros((2, feat_dim + 1))
    gstats[0, :feat_dim], gstats[0, feat_dim] = cnt * mean, cnt
    gstats[1, :feat_dim] = cnt * (std ** 2 + mean ** 2)
    means = rng.standard_normal((num_gauss, lda_dim)) * 0.7
    var = rng.uniform(0.5, 1.5, (num_gauss, lda_dim))
    w = rng.dirichlet(np.full(num_gauss, 5.0))
    M = rng.standard_normal((num_gauss, lda_dim, ivector_dim)) * 0.15
    M[:, :, 0] = means / 2.0                                     # prior_offset * M[:, :, 0] ~ the UBM means
    P = lda_dim * (lda_dim + 1) // 2
    sinv = np.zeros((num_gauss, P))
    tri = np.tril_indices(lda_dim)
    for g in range(num_gauss):
        A = rng.standard_normal((lda_dim, lda_dim)) * 0.1
        S = A @ A.T + np.diag(1.0 / var[g])
        sinv[g] = S[tri]
    return IvectorExtractionInfo(lda, gstats, w, means / var, 1.0 / var, M, sinv, 2.0, L, R, **opts)


class IvectorExtractor:
    def __init__(self, info):
        self.info = info
        self._d = info.desc()
        self._h = lib().kamd_ivector_extractor_create(C.byref(self._d))
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_ivector_extractor_destroy(self._h)
            self._h = None

    def dim(self):
        return lib().kamd_ivector_dim(self._h)

    def num_ivectors(self, num_frames):
        return lib().kamd_ivector_num_ivectors(self._h, num_frames)

    def extract_online(self, feats, state=None, return_state=False, max_remembered_frames=1000.0):
        """[ceil(T / period) x dim] for one utterance.  state: the adaptation state the speaker's previous
        utterance left (None = fresh); return_state: also the state after this utterance with LimitFrames
        applied (--max-remembered-frames), ready for the speaker's next one."""
        f = np.ascontiguousarray(feats, np.float32)
        if f.ndim != 2 or f.shape[1] != self.info.feat_dim:
            raise KamdError("features must be [frames x %d]" % self.info.feat_dim)
        n = self.num_ivectors(f.shape[0])
        out = np.zeros((n, self.dim()), np.float32)
        dp = C.POINTER(C.c_double)
        si = np.ascontiguousarray(state, np.float64) if state is not None else None
        if si is not None and si.size != self.info.state_size():
            raise KamdError("adaptation state has %d entries, expected %d" % (si.size, self.info.state_size()))
        so = np.zeros(self.info.state_size(), np.float64) if return_state else None
        r = lib().kamd_ivector_extract_online_adapt(self._h, abi.fptr(f), f.shape[0], abi.fptr(out), n,
                                                    si.ctypes.data_as(dp) if si is not None else None,
                                                    so.ctypes.data_as(dp) if so is not None else None)
        if r < 0:
            raise KamdError(lib().kamd_last_error().decode())
        if return_state:
            check(lib().kamd_ivector_state_limit_frames(self._h, so.ctypes.data_as(dp), max_remembered_frames))
            return out[:r], so
        return out[:r]

    def last_posteriors(self, frames):
        ng = self.info.num_gselect
        g, w = np.zeros((frames, ng), np.int32), np.zeros((frames, ng), np.float32)
        check(lib().kamd_ivector_last_posteriors(self._h, abi.iptr(g), abi.fptr(w), frames))
        return g, w
