"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (kaldi_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from kaldi_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_feat.cc", "orc_nnet.cc", "orc_decoder.cc")]
    srcs.append(os.path.join(_HERE, "..", "include", "kaldi_amd.h"))
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        if not all(os.path.exists(s) for s in srcs):
            if os.path.exists(so):
                return so
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp, ip, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.orc_feat_num_frames.argtypes = [C.POINTER(abi.FrameOpts), C.c_int64]
        L.orc_mfcc_compute.argtypes = [C.POINTER(abi.MfccOpts), C.c_float, fp, C.c_int64, fp, C.c_int]
        L.orc_fbank_compute.argtypes = [C.POINTER(abi.FbankOpts), C.c_float, fp, C.c_int64, fp, C.c_int]
        L.orc_nnet_context.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, ip, ip]
        L.orc_nnet_forward.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, fp,
                                       C.c_int, fp, fp, C.c_int]
        L.orc_nnet_forward_blas.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p, fp, C.c_int]
        L.orc_nnet_forward_blas_chunked.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_void_p, fp, C.c_int]
        L.orc_nnet_forward_slots.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int,
                                             C.c_int, C.c_int, fp, C.c_int]
        L.orc_nnet_forward_chunked.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int,
                                               C.c_int, C.c_int, C.c_int, fp, C.c_int]
        L.orc_nnet_forward_batch_computer.argtypes = [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, fp,
                                                      C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, ip, C.c_int, ip]
        dp = C.POINTER(C.c_double)
        L.orc_ivector_extract_online.argtypes = [C.POINTER(abi.IvectorDesc), fp, C.c_int, fp, C.c_int, fp, fp, ip, fp, ip, dp, dp]
        L.orc_ivector_extract_streaming.argtypes = [C.POINTER(abi.IvectorDesc), fp, C.c_int, ip, C.c_int, fp, dp, dp]
        L.orc_ivector_state_limit_frames.argtypes = [C.POINTER(abi.IvectorDesc), dp, C.c_float]
        L.orc_ivector_state_limit_frames.restype = None
        L.orc_linear_cgd.argtypes = [C.c_int, C.c_int, dp, dp, dp]
        L.orc_online_cmvn.argtypes = [C.POINTER(abi.IvectorDesc), fp, C.c_int, fp]
        L.orc_online_cmvn.restype = None
        L.orc_posterior_entry.argtypes = [fp, C.c_int, C.c_int, C.c_float, ip, fp, ip]
        L.orc_posterior_entry.restype = C.c_float
        L.orc_am_gmm_loglikes.argtypes = [C.c_int, C.c_int, ip, fp, fp, fp, fp, C.c_int, C.c_float, fp]
        L.orc_am_gmm_loglikes.restype = None
        L.orc_add_deltas.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_add_deltas.restype = None
        L.orc_cmvn_acc_stats.argtypes = [fp, C.c_int, C.c_int, dp]
        L.orc_cmvn_acc_stats.restype = None
        L.orc_cmvn_acc_stats_weighted.argtypes = [fp, C.c_int, C.c_int, fp, dp]
        L.orc_cmvn_acc_stats_weighted.restype = None
        L.orc_cmvn_apply.argtypes = [dp, C.c_int, fp, C.c_int, C.c_int]
        L.orc_cmvn_apply_reverse.argtypes = [dp, C.c_int, fp, C.c_int, C.c_int]
        L.orc_decoder_create.restype = C.c_void_p
        L.orc_decoder_create.argtypes = [C.c_int32, C.c_int32, i64p, C.c_void_p, fp,
                                         C.POINTER(abi.DecoderConfig), ip, C.c_int32, C.c_int]
        L.orc_graph_create.restype = C.c_void_p
        L.orc_graph_create.argtypes = [C.c_int32, C.c_int32, i64p, C.c_void_p, fp]
        L.orc_graph_destroy.argtypes = [C.c_void_p]
        L.orc_graph_destroy.restype = None
        L.orc_decoder_create_on.restype = C.c_void_p
        L.orc_decoder_create_on.argtypes = [C.c_void_p, C.POINTER(abi.DecoderConfig), ip, C.c_int32, C.c_int]
        for f in ("destroy", "init", "finalize"):
            getattr(L, "orc_decoder_" + f).argtypes = [C.c_void_p]
            getattr(L, "orc_decoder_" + f).restype = None
        L.orc_decoder_advance.argtypes = [C.c_void_p, fp, C.c_int, C.c_int]
        L.orc_decoder_advance.restype = None
        L.orc_decoder_num_frames_decoded.argtypes = [C.c_void_p]
        L.orc_decoder_num_toks.argtypes = [C.c_void_p]
        L.orc_decoder_final_relative_cost.argtypes = [C.c_void_p]
        L.orc_decoder_final_relative_cost.restype = C.c_float
        L.orc_endpoint_detected.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_trailing_silence_length.argtypes = [ip, C.c_int, ip, ip, C.c_int]
        L.orc_decoder_lattice_size.argtypes = [C.c_void_p, C.POINTER(abi.LatticeSize)]
        L.orc_decoder_get_raw_lattice.argtypes = [C.c_void_p, ip, ip, fp, fp, C.c_void_p]
        L.orc_decoder_lattice_size_ufp.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.LatticeSize)]
        L.orc_decoder_get_raw_lattice_ufp.argtypes = [C.c_void_p, C.c_int, ip, ip, fp, fp, C.c_void_p]
        L.orc_lattice_best_path.argtypes = [C.c_int, C.c_int, fp, C.c_int, C.c_void_p, ip, C.c_int,
                                            ip, ip, C.c_int, ip, fp, fp]
        L.orc_decoder_get_trace.argtypes = [C.c_void_p, ip, fp, fp, C.c_int]
        L.orc_decoder_get_counters.argtypes = [C.c_void_p, i64p]
        L.orc_decoder_get_counters.restype = None
        _LIB = L
    return _LIB


# ------------------------------------------------------------------ features
def mfcc(opts, wave, vtln_warp=1.0):
    wave = np.ascontiguousarray(wave, np.float32)
    T = lib().orc_feat_num_frames(C.byref(opts.frame), wave.size)
    out = np.zeros((max(T, 0), opts.num_ceps), np.float32)
    r = lib().orc_mfcc_compute(C.byref(opts), vtln_warp, abi.fptr(wave), wave.size,
                               abi.fptr(out), out.shape[0])
    assert r == T, r
    return out


def fbank(opts, wave, vtln_warp=1.0):
    wave = np.ascontiguousarray(wave, np.float32)
    T = lib().orc_feat_num_frames(C.byref(opts.frame), wave.size)
    dim = opts.mel.num_bins + (1 if opts.use_energy else 0)
    out = np.zeros((max(T, 0), dim), np.float32)
    r = lib().orc_fbank_compute(C.byref(opts), vtln_warp, abi.fptr(wave), wave.size,
                                abi.fptr(out), out.shape[0])
    assert r == T, r
    return out


# ---------------------------------------------------------------------- nnet
def nnet_forward(model, feats, ivector=None):
    feats = np.ascontiguousarray(feats, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    P = model.layers[-1].out_dim
    out = np.zeros((n_out, P), np.float32)
    iv = None if ivector is None else np.ascontiguousarray(ivector, np.float32)
    d = model.descs()
    r = lib().orc_nnet_forward(d, len(model.layers), model.input_dim, model.subsampling,
                               abi.fptr(feats), T, abi.fptr(iv), abi.fptr(out), n_out)
    assert r == n_out, r
    return out


_SGEMM = None


def cblas_sgemm():
    """Address of cblas_sgemm (ILP64) in the OpenBLAS that numpy ships, with OpenBLAS set to ONE thread (the callers bring
    their own threads, one utterance each, like decode.sh --nj); None when no such library is found."""
    global _SGEMM
    if _SGEMM is None:
        import glob
        _SGEMM = 0
        for path in glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so")) + \
                glob.glob(os.path.join(os.path.dirname(np.__file__), ".libs", "libscipy_openblas64_*.so")):
            try:
                L = C.CDLL(path)
                L.scipy_openblas_set_num_threads64_(1)
                _SGEMM = C.cast(L.scipy_cblas_sgemm64_, C.c_void_p).value
                cblas_sgemm.lib = L
                break
            except (OSError, AttributeError):
                continue
    return _SGEMM or None


def nnet_forward_blas(model, feats, ivector=None, frames_per_chunk=50):
    """The reference's CPU forward: DecodableNnetSimple's chunks, every Propagate an sgemm (oracle/orc_nnet_blas.cc)."""
    sg = cblas_sgemm()
    if sg is None:
        raise RuntimeError("no OpenBLAS with an ILP64 cblas_sgemm next to numpy")
    feats = np.ascontiguousarray(feats, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    out = np.zeros((n_out, model.layers[-1].out_dim), np.float32)
    iv = None if ivector is None else np.ascontiguousarray(ivector, np.float32)
    r = lib().orc_nnet_forward_blas(model.descs(), len(model.layers), model.input_dim, model.subsampling, abi.fptr(feats), T,
                                    abi.fptr(iv), int(frames_per_chunk), C.c_void_p(sg), abi.fptr(out), n_out)
    assert r == n_out, r
    return out


def nnet_forward_blas_chunked(model, feats, online_ivectors, ivector_period=10, frames_per_chunk=50):
    """nnet_forward_blas with --online-ivectors: every chunk with the row GetCurrentIvector picks for its middle (the sgemm twin of
    nnet_forward_chunked; what nnet3-latgen-faster runs on a CPU in the recipe's decode.sh)."""
    sg = cblas_sgemm()
    if sg is None:
        raise RuntimeError("no OpenBLAS with an ILP64 cblas_sgemm next to numpy")
    feats = np.ascontiguousarray(feats, np.float32)
    iv = np.ascontiguousarray(online_ivectors, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    out = np.zeros((n_out, model.layers[-1].out_dim), np.float32)
    r = lib().orc_nnet_forward_blas_chunked(model.descs(), len(model.layers), model.input_dim, model.subsampling, abi.fptr(feats), T,
                                            abi.fptr(iv), iv.shape[0], iv.shape[1], int(ivector_period), int(frames_per_chunk),
                                            C.c_void_p(sg), abi.fptr(out), n_out)
    assert r == n_out, r
    return out


def nnet_forward_slots(model, feats, slot_table, slot_first, period):
    """looped-decodable i-vectors: first-layer row t reads slot_table[floor(t / period) - slot_first] (clamped)"""
    feats = np.ascontiguousarray(feats, np.float32)
    tab = np.ascontiguousarray(slot_table, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    out = np.zeros((n_out, model.layers[-1].out_dim), np.float32)
    r = lib().orc_nnet_forward_slots(model.descs(), len(model.layers), model.input_dim, model.subsampling, abi.fptr(feats), T,
                                     abi.fptr(tab), slot_first, tab.shape[0], tab.shape[1], period, abi.fptr(out), n_out)
    assert r == n_out, r
    return out


def nnet_forward_chunked(model, feats, online_ivectors, ivector_period=10, frames_per_chunk=50):
    """DecodableNnetSimple with online ivectors: one ivector per chunk of frames_per_chunk."""
    feats = np.ascontiguousarray(feats, np.float32)
    iv = np.ascontiguousarray(online_ivectors, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    out = np.zeros((n_out, model.layers[-1].out_dim), np.float32)
    d = model.descs()
    r = lib().orc_nnet_forward_chunked(d, len(model.layers), model.input_dim, model.subsampling, abi.fptr(feats), T,
                                       abi.fptr(iv), iv.shape[0], iv.shape[1], ivector_period, frames_per_chunk,
                                       abi.fptr(out), n_out)
    assert r == n_out, r
    return out


def nnet_forward_batch_computer(model, feats, online_ivectors, ivector_period=10, frames_per_chunk=50, return_tasks=False):
    """NnetBatchComputer::SplitUtteranceIntoTasks + ComputeSimple + MergeTaskOutput (nnet3-latgen-faster-batch's forward) for one
    utterance: chunks of frames_per_chunk // subsampling output frames, the last one ending on the utterance's last frame, one
    online i-vector per task.  return_tasks: also the task table [n, 6] = first_used_output_frame_index,
    num_initial_unused_output_frames, num_used_output_frames, num_output_frames, first_input_t, i-vector row."""
    feats = np.ascontiguousarray(feats, np.float32)
    iv = None if online_ivectors is None else np.ascontiguousarray(online_ivectors, np.float32)
    T = feats.shape[0]
    n_out = (T + model.subsampling - 1) // model.subsampling
    out = np.zeros((n_out, model.layers[-1].out_dim), np.float32)
    cap = n_out + 2
    tasks = np.zeros((cap, 6), np.int32)
    n_tasks = C.c_int(0)
    left, right = model.context()
    r = lib().orc_nnet_forward_batch_computer(model.descs(), len(model.layers), model.input_dim, model.subsampling, int(left), int(right),
                                              abi.fptr(feats), T, abi.fptr(iv) if iv is not None else None,
                                              0 if iv is None else iv.shape[0], 0 if iv is None else iv.shape[1], int(ivector_period),
                                              int(frames_per_chunk), abi.fptr(out), n_out, abi.iptr(tasks), cap, C.byref(n_tasks))
    assert r == n_out, r
    return (out, tasks[:n_tasks.value].copy()) if return_tasks else out


def ivector_extract_online(info, feats, diagnostics=False, state=None, return_state=False, max_remembered_frames=1000.0):
    """ivector-extract-online2 for one utterance: [ceil(T/period) x dim].  state = the adaptation state
    left by the speaker's previous utterance (None: fresh); return_state: also the state after this
    one, LimitFrames applied, as the binary carries it to the speaker's next utterance."""
    f = np.ascontiguousarray(feats, np.float32)
    T = f.shape[0]
    d = info.desc()
    n = (T + info.ivector_period - 1) // info.ivector_period
    out = np.zeros((n, info.ivector_dim), np.float32)
    D, ng = info.lda.shape[0], info.num_gselect
    nl, rl = np.zeros((T, D), np.float32), np.zeros((T, D), np.float32)
    pg, pw = np.zeros((T, ng), np.int32), np.zeros((T, ng), np.float32)
    worse = C.c_int32()
    dp = C.POINTER(C.c_double)
    ss = info.state_size()
    st_in = np.ascontiguousarray(state, np.float64) if state is not None else None
    st_out = np.zeros(ss, np.float64)
    r = lib().orc_ivector_extract_online(C.byref(d), abi.fptr(f), T, abi.fptr(out), n, abi.fptr(nl), abi.fptr(rl), abi.iptr(pg),
                                         abi.fptr(pw), C.byref(worse), st_in.ctypes.data_as(dp) if st_in is not None else None,
                                         st_out.ctypes.data_as(dp))
    assert r == n, r
    if return_state:
        lib().orc_ivector_state_limit_frames(C.byref(d), st_out.ctypes.data_as(dp), max_remembered_frames)
    if diagnostics:
        res = out, dict(norm_lda=nl, raw_lda=rl, post_gauss=pg, post_weight=pw, cg_got_worse=worse.value)
    else:
        res = out
    return (res, st_out) if return_state else res


def ivector_extract_streaming(info, feats, upto, state=None):
    """OnlineIvectorFeature::GetFrame(upto[c] - 1) for c = 0, 1, ... with use_most_recent_ivector: [n_calls x dim]"""
    f = np.ascontiguousarray(feats, np.float32)
    u = np.ascontiguousarray(upto, np.int32)
    d = info.desc()
    out = np.zeros((u.size, info.ivector_dim), np.float32)
    dp = C.POINTER(C.c_double)
    st_in = np.ascontiguousarray(state, np.float64) if state is not None else None
    st_out = np.zeros(info.state_size(), np.float64)
    r = lib().orc_ivector_extract_streaming(C.byref(d), abi.fptr(f), f.shape[0], abi.iptr(u), u.size, abi.fptr(out),
                                            st_in.ctypes.data_as(dp) if st_in is not None else None, st_out.ctypes.data_as(dp))
    assert r == u.size, r
    return out, st_out


def ivector_extract_streaming_weighted(info, feats, upto, lists, state=None):
    """UpdateStatsUntilFrameWeighted(upto[c] - 1) for c = 0, 1, ...; lists[c] = the merged (frame, weight) pairs of call c
    (DeltaWeightQueue.pop_until): [n_calls x dim], final adaptation state"""
    f = np.ascontiguousarray(feats, np.float32)
    u = np.ascontiguousarray(upto, np.int32)
    off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
    fr = np.ascontiguousarray([p[0] for l in lists for p in l] or [0], np.int32)
    wt = np.ascontiguousarray([p[1] for l in lists for p in l] or [0], np.float32)
    d = info.desc()
    out = np.zeros((u.size, info.ivector_dim), np.float32)
    dp = C.POINTER(C.c_double)
    st_in = np.ascontiguousarray(state, np.float64) if state is not None else None
    st_out = np.zeros(info.state_size(), np.float64)
    r = lib().orc_ivector_extract_streaming_weighted(C.byref(d), abi.fptr(f), f.shape[0], abi.iptr(u), u.size, abi.iptr(off), abi.iptr(fr),
                                                     abi.fptr(wt), abi.fptr(out), st_in.ctypes.data_as(dp) if st_in is not None else None,
                                                     st_out.ctypes.data_as(dp))
    assert r == u.size, r
    return out, st_out


class OnlineSilenceWeighting:
    """online2/online-ivector-feature.{h:453-535, cc:447-668}, statement by statement.  A token is whatever hashable
    the caller's traceback names it by (the reference compares Token pointers)."""

    def __init__(self, tid2phone, silence_phones, silence_weight, max_state_duration=-1.0, frame_subsampling_factor=1):
        self.tid2phone = np.asarray(tid2phone)
        self.silence_phones = set(int(p) for p in silence_phones)
        self.silence_weight = np.float32(silence_weight)
        self.max_state_duration = max_state_duration
        self.fs = int(frame_subsampling_factor)
        assert self.fs >= 1
        self.frame_info = []                     # [token, transition_id, current_weight]
        self.num_frames_output_and_correct = 0

    def Active(self):
        return len(self.silence_phones) > 0 and self.silence_weight != 1.0

    def _resize(self, n):
        while len(self.frame_info) < n:
            self.frame_info.append([None, -1, np.float32(0.0)])

    def ComputeCurrentTraceback(self, num_frames_decoded, path):
        """path: (transition_id, token) of frames num_frames_decoded-1, num_frames_decoded-2, ... (BestPathEnd without
        final-probs + TraceBackBestPath, input-epsilon arcs skipped)  (.cc:464-510)"""
        num_frames_prev = len(self.frame_info)
        if num_frames_prev < num_frames_decoded:
            self._resize(num_frames_decoded)
        if num_frames_prev > num_frames_decoded and self.frame_info[num_frames_decoded][1] != -1:
            raise RuntimeError("Number of frames decoded decreased")
        if num_frames_decoded == 0:
            return
        frame = num_frames_decoded - 1
        it = iter(path)
        while frame >= 0:
            tid, tok = next(it)
            if self.frame_info[frame][0] is not None and self.frame_info[frame][0] == tok:
                break
            if self.num_frames_output_and_correct > frame:
                self.num_frames_output_and_correct = frame
            self.frame_info[frame][0] = tok
            self.frame_info[frame][1] = int(tid)
            frame -= 1

    def GetBeginFrame(self):                    # .cc:521-571
        max_duration = int(self.max_state_duration)
        if max_duration <= 0 or self.num_frames_output_and_correct == 0:
            return self.num_frames_output_and_correct
        t_last_untouched = self.num_frames_output_and_correct - 1
        t_end = len(self.frame_info)
        transition_id = self.frame_info[t_last_untouched][1]
        lower_search_bound = max(0, t_last_untouched - max_duration)
        upper_search_bound = min(t_last_untouched + max_duration, t_end - 1)
        t_lower = t_last_untouched
        while t_lower > lower_search_bound and self.frame_info[t_lower - 1][1] == transition_id:
            t_lower -= 1
        t_upper = t_last_untouched
        while t_upper < upper_search_bound and self.frame_info[t_upper + 1][1] == transition_id:
            t_upper += 1
        run_length = t_upper - t_lower + 1
        if run_length <= max_duration:
            return self.num_frames_output_and_correct
        old_run_length = t_last_untouched - t_lower + 1
        if old_run_length > max_duration:
            ans = t_upper - max_duration
            assert ans >= t_lower
            return ans
        return t_lower

    def GetDeltaWeights(self, num_frames_ready_in):
        """-> [(input frame, delta weight)]  (.cc:573-668).  Note: no statement of the reference raises
        num_frames_output_and_correct_ (it starts at 0 and ComputeCurrentTraceback only lowers it), so begin_frame is 0."""
        fs = self.fs
        num_frames_ready = (num_frames_ready_in + fs - 1) // fs
        max_state_duration = int(self.max_state_duration)
        silence_weight = self.silence_weight
        delta_weights = []
        if len(self.frame_info) < num_frames_ready:
            self._resize(num_frames_ready)
        begin_frame = self.GetBeginFrame()
        frames_out = len(self.frame_info) - begin_frame
        assert frames_out >= 0
        frame_weight = [np.float32(1.0)] * frames_out
        if frames_out == 0:
            return delta_weights
        if self.frame_info[begin_frame][1] == -1:
            weight = silence_weight if begin_frame == 0 else self.frame_info[begin_frame - 1][2]
            frame_weight = [weight] * frames_out
        else:
            current_run_start_offset = 0
            for offset in range(frames_out):
                frame = begin_frame + offset
                transition_id = self.frame_info[frame][1]
                if transition_id == -1:
                    frame_weight[offset] = frame_weight[offset - 1]
                else:
                    phone = int(self.tid2phone[transition_id])
                    if phone in self.silence_phones:
                        frame_weight[offset] = silence_weight
                    if max_state_duration > 0 and (offset + 1 == frames_out or transition_id != self.frame_info[frame + 1][1]):
                        run_length = offset - current_run_start_offset + 1
                        if run_length >= max_state_duration:
                            for offset2 in range(current_run_start_offset, offset + 1):
                                frame_weight[offset2] = silence_weight
                        if offset + 1 < frames_out:
                            current_run_start_offset = offset + 1
        for offset in range(frames_out):
            frame = begin_frame + offset
            old_weight = self.frame_info[frame][2]
            new_weight = np.float32(frame_weight[offset])
            weight_diff = np.float32(new_weight - old_weight)
            self.frame_info[frame][2] = new_weight
            if weight_diff != 0.0 or offset + 1 == frames_out:
                for i in range(fs):
                    delta_weights.append((frame * fs + i, weight_diff))
        return delta_weights


class DeltaWeightQueue:
    """OnlineIvectorFeature's delta_weights_ priority queue: UpdateFrameWeights (.cc:159-174) and what
    UpdateStatsUntilFrameWeighted(frame) (.cc:263-306) pops and hands to UpdateStatsForFrames, after its
    MergePairVectorSumming (util/stl-utils.h:290-315)."""

    def __init__(self):
        import heapq
        self._hq = heapq
        self.heap = []
        self.most_recent_frame_with_weight = -1

    def UpdateFrameWeights(self, delta_weights):
        for frame, w in delta_weights:
            assert frame >= 0
            self._hq.heappush(self.heap, (int(frame), float(np.float32(w))))
            self.most_recent_frame_with_weight = max(self.most_recent_frame_with_weight, int(frame))

    def pop_until(self, frame):
        assert frame <= self.most_recent_frame_with_weight
        popped = []
        while self.heap and self.heap[0][0] <= frame:
            popped.append(self._hq.heappop(self.heap))
        merged = []
        for fr, w in popped:                     # sorted by frame already
            if merged and merged[-1][0] == fr:
                merged[-1][1] = np.float32(merged[-1][1] + np.float32(w))
            else:
                merged.append([fr, np.float32(w)])
        return [(fr, float(w)) for fr, w in merged if w != 0.0]


def linear_cgd(A_packed, b, x0, max_iters):
    A = np.ascontiguousarray(A_packed, np.float64)
    bb = np.ascontiguousarray(b, np.float64)
    x = np.array(x0, np.float64)
    dp = C.POINTER(C.c_double)
    k = lib().orc_linear_cgd(max_iters, bb.size, A.ctypes.data_as(dp), bb.ctypes.data_as(dp), x.ctypes.data_as(dp))
    return x, k


def online_cmvn(info, feats):
    f = np.ascontiguousarray(feats, np.float32)
    out = np.zeros_like(f)
    d = info.desc()
    lib().orc_online_cmvn(C.byref(d), abi.fptr(f), f.shape[0], abi.fptr(out))
    return out


def posterior_entry(loglikes, num_gselect, min_post):
    ll = np.ascontiguousarray(loglikes, np.float32)
    g, p = np.zeros(ll.size, np.int32), np.zeros(ll.size, np.float32)
    n = C.c_int32()
    r = lib().orc_posterior_entry(abi.fptr(ll), ll.size, num_gselect, min_post, abi.iptr(g), abi.fptr(p), C.byref(n))
    return r, g[:n.value], p[:n.value]


def am_gmm_loglikes(am, feats, scale=1.0):
    """DecodableAmDiagGmmScaled's matrix [T x num_pdfs] for a kaldi_amd.gmm.AmDiagGmm"""
    f = np.ascontiguousarray(feats, np.float32)
    out = np.zeros((f.shape[0], am.num_pdfs), np.float32)
    lib().orc_am_gmm_loglikes(am.num_pdfs, am.dim, abi.iptr(am.mix_off), abi.fptr(am.gconsts), abi.fptr(am.means_invvars),
                              abi.fptr(am.inv_vars), abi.fptr(f), f.shape[0], scale, abi.fptr(out))
    return out


def add_deltas(feats, order=2, window=2):
    f = np.ascontiguousarray(feats, np.float32)
    out = np.zeros((f.shape[0], (order + 1) * f.shape[1]), np.float32)
    lib().orc_add_deltas(abi.fptr(f), f.shape[0], f.shape[1], order, window, abi.fptr(out))
    return out


def cmvn_acc_stats(feats, stats=None, weights=None):
    f = np.ascontiguousarray(feats, np.float32)
    st = np.zeros((2, f.shape[1] + 1), np.float64) if stats is None else np.array(stats, np.float64)
    if weights is not None:
        w = np.ascontiguousarray(weights, np.float32)
        assert w.size == f.shape[0]
        lib().orc_cmvn_acc_stats_weighted(abi.fptr(f), f.shape[0], f.shape[1], abi.fptr(w), st.ctypes.data_as(C.POINTER(C.c_double)))
        return st
    lib().orc_cmvn_acc_stats(abi.fptr(f), f.shape[0], f.shape[1], st.ctypes.data_as(C.POINTER(C.c_double)))
    return st


def cmvn_apply(feats, stats, norm_vars=False, reverse=False):
    f = np.array(feats, np.float32)
    st = np.ascontiguousarray(stats, np.float64)
    r = (lib().orc_cmvn_apply_reverse if reverse else lib().orc_cmvn_apply)(st.ctypes.data_as(C.POINTER(C.c_double)), int(norm_vars), abi.fptr(f), f.shape[0], f.shape[1])
    assert r == 0, "Insufficient stats"
    return f


def nnet_context(model):
    l, r = C.c_int32(), C.c_int32()
    lib().orc_nnet_context(model.descs(), len(model.layers), C.byref(l), C.byref(r))
    return l.value, r.value


# ------------------------------------------------------------------- decoder
class Lattice:
    """Raw lattice in canonical numbering (see include/kaldi_amd.h)."""

    def __init__(self, start, frame, hclg, cost, final, arcs, num_frames):
        self.start, self.frame, self.hclg, self.cost = start, frame, hclg, cost
        self.final, self.arcs, self.num_frames = final, arcs, num_frames

    def best_path(self):
        n, m = self.frame.size, self.arcs.size
        ali = np.zeros(max(m, 1), np.int32)
        words = np.zeros(max(m, 1), np.int32)
        na, nw = C.c_int(), C.c_int()
        g, a = C.c_float(), C.c_float()
        arcs = np.ascontiguousarray(self.arcs)
        r = lib().orc_lattice_best_path(n, self.start, abi.fptr(self.final), m,
                                        arcs.ctypes.data_as(C.c_void_p),
                                        abi.iptr(ali), ali.size, C.byref(na), abi.iptr(words),
                                        words.size, C.byref(nw), C.byref(g), C.byref(a))
        if r != 0:
            return None
        return dict(alignment=ali[:na.value].copy(), words=words[:nw.value].copy(),
                    graph_cost=g.value, acoustic_cost=a.value)


    def best_path_frames(self):
        """(transition_id, HCLG state of the token the arc leaves) per decoded frame, newest first: what
        OnlineSilenceWeighting::ComputeCurrentTraceback collects with BestPathEnd + TraceBackBestPath, skipping
        input-epsilon arcs (online2/online-ivector-feature.cc:478-505).  None when no path exists."""
        n, m = self.frame.size, self.arcs.size
        arcs = np.ascontiguousarray(self.arcs)
        path = np.zeros(max(m, 1), np.int32)
        k = C.c_int()
        r = lib().orc_lattice_best_path_arcs(n, self.start, abi.fptr(self.final), m, arcs.ctypes.data_as(C.c_void_p), abi.iptr(path),
                                             path.size, C.byref(k))
        if r != 0:
            return None
        out = []
        for i in path[:k.value][::-1]:
            a = arcs[int(i)]
            if a["ilabel"] != 0:
                out.append((int(a["ilabel"]), int(self.hclg[int(a["src"])])))
        return out


class _Graph:
    """Host copy of a decoding graph in the oracle's layout, shared read-only by its decoders (the reference
    hands one const fst::Fst& to every decoder object)."""

    def __init__(self, g):
        arcs = np.ascontiguousarray(g.arcs)
        self._h = lib().orc_graph_create(g.num_states, g.start, abi.iptr(np.ascontiguousarray(g.arc_off, np.int64), C.c_int64),
                                         arcs.ctypes.data_as(C.c_void_p), abi.fptr(np.ascontiguousarray(g.final, np.float32)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_graph_destroy(self._h)
            self._h = None


_graph_lock = __import__("threading").Lock()


def shared_graph(g):
    with _graph_lock:
        sg = getattr(g, "_orc_graph", None)
        if sg is None:
            sg = _Graph(g)
            try:
                g._orc_graph = sg
            except AttributeError:           # an object that takes no attributes: a private copy then
                pass
        return sg


class Decoder:
    """orc_decoder: mode 0 = faithful to the reference's order, 1 = canonical."""

    def __init__(self, g, cfg, mode=1):
        self.g, self.cfg, self.mode = g, cfg, mode
        self._graph = shared_graph(g)          # one host copy per Hclg object, shared by all its decoders
        self._h = lib().orc_decoder_create_on(self._graph._h, C.byref(cfg), abi.iptr(np.ascontiguousarray(g.tid2pdf, np.int32)),
                                              g.tid2pdf.size - 1, mode)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_decoder_destroy(self._h)
            self._h = None

    def InitDecoding(self):
        lib().orc_decoder_init(self._h)

    def AdvanceDecoding(self, loglikes):
        ll = np.ascontiguousarray(loglikes, np.float32)
        lib().orc_decoder_advance(self._h, abi.fptr(ll), ll.shape[1], ll.shape[0])

    def FinalizeDecoding(self):
        lib().orc_decoder_finalize(self._h)

    def Decode(self, loglikes):
        self.InitDecoding()
        self.AdvanceDecoding(loglikes)
        self.FinalizeDecoding()

    def NumFramesDecoded(self):
        return lib().orc_decoder_num_frames_decoded(self._h)

    def FinalRelativeCost(self):
        return lib().orc_decoder_final_relative_cost(self._h)

    def GetRawLattice(self, use_final_probs=True):
        """lattice-faster-decoder.cc:113-196, also on a live decoder (final costs computed on the spot; use_final_probs
        False -- legal only before FinalizeDecoding -- makes every token of the last frame final with weight One)."""
        sz = abi.LatticeSize()
        rc = lib().orc_decoder_lattice_size_ufp(self._h, int(bool(use_final_probs)), C.byref(sz))
        if rc == -2:
            raise ValueError("You cannot call FinalizeDecoding() and then call GetRawLattice() with use_final_probs == false")
        if rc != 0:
            return None
        n, m = sz.num_states, sz.num_arcs
        fr, hc = np.zeros(n, np.int32), np.zeros(n, np.int32)
        co, fi = np.zeros(n, np.float32), np.zeros(n, np.float32)
        arcs = np.zeros(m, abi.LAT_ARC_DTYPE)
        lib().orc_decoder_get_raw_lattice_ufp(self._h, int(bool(use_final_probs)), abi.iptr(fr), abi.iptr(hc), abi.fptr(co),
                                              abi.fptr(fi), arcs.ctypes.data_as(C.c_void_p))
        return Lattice(sz.start, fr, hc, co, fi, arcs, sz.num_frames)

    def trace(self):
        n = self.NumFramesDecoded()
        nt, cu, of = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        k = lib().orc_decoder_get_trace(self._h, abi.iptr(nt), abi.fptr(cu), abi.fptr(of), n)
        return nt[:k], cu[:k], of[:k]

    def counters(self):
        c = np.zeros(8, np.int64)
        lib().orc_decoder_get_counters(self._h, abi.iptr(c, C.c_int64))
        return c


def endpoint_detected(rules, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds, final_relative_cost):
    """rules: 5 x (must_contain_nonsilence, min_trailing_silence, max_relative_cost, min_utterance_length)"""
    r = np.ascontiguousarray(rules, np.float32).reshape(5, 4)
    return bool(lib().orc_endpoint_detected(abi.fptr(r), int(num_frames_decoded), int(trailing_silence_frames),
                                            float(frame_shift_in_seconds), float(final_relative_cost)))


def trailing_silence_length(alignment, tid2phone, silence_phones):
    a = np.ascontiguousarray(alignment, np.int32)
    tp = np.ascontiguousarray(tid2phone, np.int32)
    sp = np.ascontiguousarray(list(silence_phones), np.int32)
    return lib().orc_trailing_silence_length(abi.iptr(a), a.size, abi.iptr(tp), abi.iptr(sp), sp.size)
