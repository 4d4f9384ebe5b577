"""Streaming decode (BASELINE config 5), mirroring online2's classes over the C-ABI:

  OnlineMfcc                 feat/online-feature.h:78 OnlineGenericBaseFeature<MfccComputer>
  SingleUtteranceNnet3Decoder online2/online-nnet3-decoding.h:52 (feature pipeline +
                             DecodableAmNnetLoopedOnline + LatticeFasterOnlineDecoder)
as driven by online2bin/online2-wav-nnet3-latgen-faster.cc:211-285:
  AcceptWaveform(chunk) -> AdvanceDecoding() ... InputFinished() -> AdvanceDecoding() ->
  FinalizeDecoding() -> GetLattice / GetBestPath.
Everything stays on the device between the calls (chunked features on-device)."""
import ctypes as C

import numpy as np

from . import abi, decoder, feat
from ._lib import KamdError, check, lib


class OnlineMfcc:
    def __init__(self, opts=None):
        self.computer = feat.Mfcc(opts)
        self._h = lib().kamd_online_feat_create(self.computer._h)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_online_feat_destroy(self._h)
            self._h = None

    def Dim(self):
        return self.computer.Dim()

    def AcceptWaveform(self, sampling_rate, waveform):
        w = np.ascontiguousarray(waveform, np.float32)
        check(lib().kamd_online_feat_accept_waveform(self._h, float(sampling_rate), abi.fptr(w), w.size))

    def InputFinished(self):
        check(lib().kamd_online_feat_input_finished(self._h))

    def NumFramesReady(self):
        return lib().kamd_online_feat_num_frames_ready(self._h)

    def IsLastFrame(self, frame):
        return bool(lib().kamd_online_feat_is_last_frame(self._h, frame))

    def GetFrames(self, first, n):
        out = np.zeros((n, self.Dim()), np.float32)
        check(lib().kamd_online_feat_get_frames(self._h, first, n, abi.fptr(out)))
        return out

    def GetFrame(self, frame):
        return self.GetFrames(frame, 1)[0]

    def device_frames(self):
        ld = C.c_int32()
        p = lib().kamd_online_feat_device_frames(self._h, C.byref(ld))
        return p, ld.value


class SingleUtteranceNnet3Decoder:
    def __init__(self, mfcc_opts, nnet, graph, config, sizes=None, max_seconds=40.0):
        """nnet: decoder.Nnet, graph: decoder.Graph (shared, read-only), config: DecoderConfig."""
        self.features = OnlineMfcc(mfcc_opts)
        self.nnet, self.graph = nnet, graph
        sub = lib().kamd_nnet_frame_subsampling_factor(nnet._h)
        if sizes is None:
            from .pipeline import default_sizes
            sizes = default_sizes(config, 1, int(max_seconds * 100 / sub) + 2)
        self.decoder = decoder.LatticeFasterDecoder(graph, config, sizes)
        self.P = nnet.OutputDim()
        self._finished = False
        self._ll = None
        self._ll_rows = 0
        self.decoder.InitDecoding()          # online-nnet3-decoding.cc:40

    def AcceptWaveform(self, sampling_rate, waveform):
        self.features.AcceptWaveform(sampling_rate, waveform)

    def InputFinished(self):
        self.features.InputFinished()
        self._finished = True

    def NumFramesReady(self):
        return lib().kamd_nnet_num_frames_ready(self.nnet._h, self.features.NumFramesReady(), int(self._finished))

    def NumFramesDecoded(self):
        return self.decoder.NumFramesDecoded()

    def AdvanceDecoding(self):
        """Decode every output frame that became ready (online-nnet3-decoding.cc:51-53)."""
        done, ready = self.decoder.NumFramesDecoded(), self.NumFramesReady()
        n = ready - done
        if n <= 0:
            return 0
        nbytes = n * self.P * 4
        d_ll = lib().kamd_malloc(nbytes)
        if not d_ll:
            raise KamdError(lib().kamd_last_error().decode())
        try:
            p, ld = self.features.device_frames()
            check(lib().kamd_nnet_forward_range(self.nnet._h, p, ld, self.features.NumFramesReady(),
                                                int(self._finished), done, n, d_ll, self.P))
            task = abi.DecodeTask(0, n, d_ll, self.P, 0)
            check(lib().kamd_decoder_advance(self.decoder._dec, C.byref(task), 1, None))
            check(lib().kamd_decoder_sync(self.decoder._dec))
            if self._ll is not None:       # keep a host copy for tests (loglikes())
                buf = np.zeros((n, self.P), np.float32)
                check(lib().kamd_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), d_ll, nbytes))
                self._ll.append(buf)
        finally:
            lib().kamd_free(d_ll)
        return n

    def record_loglikes(self):
        self._ll = []

    def loglikes(self):
        return np.concatenate(self._ll) if self._ll else np.zeros((0, self.P), np.float32)

    def FinalizeDecoding(self):
        self.decoder.FinalizeDecoding()

    def GetBestPath(self, end_of_utterance=True):
        """online-nnet3-decoding.cc:81-85: use_final_probs = end_of_utterance."""
        return self.decoder.GetBestPath(use_final_probs=end_of_utterance)

    def GetRawLattice(self, use_final_probs=True):
        return self.decoder.GetRawLattice(use_final_probs)

    def GetLattice(self, end_of_utterance, tid_phone=None, det_opts=None):
        """SingleUtteranceNnet3DecoderTpl::GetLattice (online2/online-nnet3-decoding.cc:66-79): the raw lattice -- of the LIVE
        decoder when FinalizeDecoding has not been called, final-probs only at the end of the utterance -- through
        DeterminizeLatticePhonePrunedWrapper at the decoder's lattice beam.  -> kaldi_amd.io.CompactLattice or None."""
        from . import io as kio
        raw = self.decoder.GetRawLattice(use_final_probs=bool(end_of_utterance))
        if raw is None:
            return None
        return kio.determinize_lattice(raw, self.decoder.config.lattice_beam, tid_phone, det_opts)

    # ---- endpointing (online2/online-nnet3-decoding.cc:88-97, online2/online-endpoint.cc:71-121)
    def TrailingSilenceLength(self, tid2phone, silence_phones):
        """Frames of silence at the end of the current best path (BestPathEnd without final-probs, traced back on
        the device until the first non-silence phone).  tid2phone[tid] = TransitionModel::TransitionIdToPhone(tid)."""
        _set_silence(self.decoder, tid2phone, silence_phones)
        return int(decoder.trailing_silence_frames(self.decoder._dec, [self.decoder.lane])[0])

    def EndpointDetected(self, config, tid2phone, silence_phones, frame_shift_in_seconds=None):
        """frame_shift_in_seconds: of a DECODED frame (feature shift x frame-subsampling-factor,
        online-nnet3-decoding.cc:93-95)."""
        if frame_shift_in_seconds is None:
            frame_shift_in_seconds = 0.01 * lib().kamd_nnet_frame_subsampling_factor(self.nnet._h)
        _set_silence(self.decoder, tid2phone, silence_phones)
        det, _ = decoder.endpoint_detected(self.decoder._dec, config.to_c(), [self.decoder.lane], frame_shift_in_seconds)
        return bool(det[0])


def _set_silence(dec, tid2phone, silence_phones):
    """kamd_decoder_set_silence_phones once per (table, phone list)"""
    key = (id(tid2phone), tuple(int(p) for p in silence_phones))
    if getattr(dec, "_silence_key", None) != key:
        decoder.set_silence_phones(dec._dec, tid2phone, key[1])
        dec._silence_key = key
        dec._silence_tbl = tid2phone           # keeps id() stable


class OnlineEndpointRule:
    """online2/online-endpoint.h:113-143."""

    def __init__(self, must_contain_nonsilence=True, min_trailing_silence=1.0, max_relative_cost=float("inf"),
                 min_utterance_length=0.0):
        self.must_contain_nonsilence = must_contain_nonsilence
        self.min_trailing_silence = min_trailing_silence
        self.max_relative_cost = max_relative_cost
        self.min_utterance_length = min_utterance_length


class OnlineEndpointConfig:
    """online2/online-endpoint.h:145-170, same five default rules."""

    def __init__(self):
        inf = float("inf")
        self.rule1 = OnlineEndpointRule(False, 5.0, inf, 0.0)
        self.rule2 = OnlineEndpointRule(True, 0.5, 2.0, 0.0)
        self.rule3 = OnlineEndpointRule(True, 1.0, 8.0, 0.0)
        self.rule4 = OnlineEndpointRule(True, 2.0, inf, 0.0)
        self.rule5 = OnlineEndpointRule(False, 0.0, inf, 20.0)

    def to_c(self):
        c = abi.EndpointConfig()
        for i, r in enumerate((self.rule1, self.rule2, self.rule3, self.rule4, self.rule5)):
            c.rule[i] = abi.EndpointRule(int(bool(r.must_contain_nonsilence)), r.min_trailing_silence, r.max_relative_cost,
                                         r.min_utterance_length)
        return c


def endpoint_detected(config, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds,
                      final_relative_cost):
    """EndpointDetected on plain numbers (online2/online-endpoint.cc:46-68): kamd_endpoint_detected, BaseFloat
    arithmetic like the reference's."""
    return decoder.endpoint_detected_from(config.to_c(), num_frames_decoded, trailing_silence_frames,
                                          frame_shift_in_seconds, final_relative_cost)


def trailing_silence_length(best_path, tid2phone, silence_phones):
    """TrailingSilenceLength (online2/online-endpoint.cc:71-102) restated on a best-path alignment (host; the
    decoders use kamd_decoder_trailing_silence_frames, which stops walking at the first non-silence frame)."""
    if best_path is None:
        return 0
    sil = set(int(p) for p in silence_phones)
    n = 0
    for tid in reversed(best_path["alignment"].tolist()):
        if int(tid2phone[tid]) in sil:
            n += 1
        else:
            break
    return n


class OnlineSilenceWeightingConfig:
    """online2/online-ivector-feature.h:404-451; registered with the prefix "ivector-silence-weighting"
    (online-nnet2-feature-pipeline.h:89-110)."""

    def __init__(self, silence_phones_str="", silence_weight=1.0, max_state_duration=-1.0):
        self.silence_phones_str, self.silence_weight, self.max_state_duration = silence_phones_str, silence_weight, max_state_duration

    def Active(self):
        return bool(self.silence_phones_str) and self.silence_weight != 1.0

    def silence_phones(self):
        """SplitStringToIntegers(silence_phones_str, ":,", false)"""
        return [int(x) for x in self.silence_phones_str.replace(",", ":").split(":") if x != ""]

    def tid_is_silence(self, tid2phone):
        sil = np.asarray(self.silence_phones(), np.int64)
        return np.ascontiguousarray(np.isin(np.asarray(tid2phone), sil), np.uint8)

    @staticmethod
    def register(po, prefix="ivector-silence-weighting"):
        po.register(prefix + ".silence-phones", str, "", "(RE weighting in iVector estimation for online decoding) List of integer ids of "
                    "silence phones, separated by colons (or commas).  Data that (according to the traceback of the decoder) corresponds to "
                    "these phones will be downweighted by --silence-weight.")
        po.register(prefix + ".silence-weight", float, 1.0, "(RE weighting in iVector estimation for online decoding) Weighting factor for "
                    "frames that the decoder trace-back identifies as silence; only relevant if the --silence-phones option is set.")
        po.register(prefix + ".max-state-duration", float, -1.0, "(RE weighting in iVector estimation for online decoding) Maximum allowed "
                    "duration of a single transition-id; runs with durations longer than this will be weighted down to the silence-weight.")

    @classmethod
    def from_options(cls, po, prefix="ivector-silence-weighting"):
        return cls(po[prefix + ".silence-phones"], po[prefix + ".silence-weight"], po[prefix + ".max-state-duration"])


class OnlineSilenceWeighting:
    """online2/online-ivector-feature.h:453-535 over kamd_silence_weighting_*: one per utterance.  The delta weights are
    queued inside (OnlineIvectorFeature::UpdateFrameWeights) and come back out through pop_until when the estimate is
    advanced (UpdateStatsUntilFrameWeighted)."""

    def __init__(self, tid2phone, config, frame_subsampling_factor=1):
        self.config = config
        t = config.tid_is_silence(tid2phone)
        self._h = lib().kamd_silence_weighting_create(t.ctypes.data_as(C.POINTER(C.c_uint8)), t.size, config.silence_weight,
                                                      config.max_state_duration, int(frame_subsampling_factor))
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_silence_weighting_destroy(self._h)
            self._h = None

    def Active(self):
        return self.config.Active()

    def ComputeCurrentTraceback(self, num_frames_decoded, tids, tokens):
        """tids / tokens: the decoder's best path without final-probs, newest frame first (decoder.frame_tracebacks)"""
        t, k = np.ascontiguousarray(tids, np.int32), np.ascontiguousarray(tokens, np.int32)
        check(lib().kamd_silence_weighting_compute_traceback(self._h, int(num_frames_decoded), abi.iptr(t), abi.iptr(k), t.size))

    def GetDeltaWeights(self, num_frames_ready_in):
        n = C.c_int32()
        check(lib().kamd_silence_weighting_get_delta_weights(self._h, int(num_frames_ready_in), C.byref(n)))
        return n.value

    def pop_until(self, frame):
        cap = lib().kamd_silence_weighting_num_pending(self._h) + 1
        fr, wt = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = C.c_int32()
        check(lib().kamd_silence_weighting_pop_until(self._h, int(frame), abi.iptr(fr), abi.fptr(wt), cap, C.byref(n)))
        return [(int(fr[i]), float(wt[i])) for i in range(n.value)]


class StreamBatch:
    """N concurrent streams decoded together (kamd_stream_batch_*): stream s = decoder lane s."""

    def __init__(self, mfcc_opts, nnet, graph, config, max_streams, max_seconds=40.0, sizes=None):
        featmod = __import__("kaldi_amd.feat", fromlist=["Mfcc"])
        self.feat = featmod.Fbank(mfcc_opts) if isinstance(mfcc_opts, abi.FbankOpts) else featmod.Mfcc(mfcc_opts)
        self.nnet, self.graph, self.config = nnet, graph, config
        sub = lib().kamd_nnet_frame_subsampling_factor(nnet._h)
        if sizes is None:
            from .pipeline import default_sizes
            sizes = default_sizes(config, max_streams, int(max_seconds * 100 / sub) + 2)
        self.dec = decoder.BatchDecoder(graph, config, sizes)
        self._h = lib().kamd_stream_batch_create(self.feat._h, nnet._h, self.dec._dec, max_streams, max_seconds,
                                                 mfcc_opts.frame.samp_freq)
        if not self._h:
            raise KamdError(lib().kamd_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kamd_stream_batch_destroy(self._h)
            self._h = None

    def set_ivector_extractor(self, extractor, frames_per_chunk=20):
        """Online i-vectors as online2-wav-nnet3-latgen-faster computes them (OnlineIvectorFeature with
        use_most_recent_ivector + DecodableNnetLoopedOnline's chunk schedule); before the first start()."""
        check(lib().kamd_stream_batch_set_ivector_extractor(self._h, extractor._h, frames_per_chunk, extractor.info.splice_right))
        sub = lib().kamd_nnet_frame_subsampling_factor(self.nnet._h)
        self.extractor, self.frames_per_chunk = extractor, sub * ((frames_per_chunk + sub - 1) // sub)   # GetChunkSize rounding

    def set_silence_weighting(self, config, tid2phone):
        """--ivector-silence-weighting.* : after set_ivector_extractor, before the first start()"""
        t = config.tid_is_silence(tid2phone)
        check(lib().kamd_stream_batch_set_silence_weighting(self._h, t.ctypes.data_as(C.POINTER(C.c_uint8)), t.size,
                                                            config.silence_weight if config.Active() else 1.0, config.max_state_duration))

    def set_compaction(self, fraction):
        """PruneActiveTokens as arena compaction when a stream's arena is fuller than `fraction` (default 0.5; 0 = never)"""
        check(lib().kamd_stream_batch_set_compaction(self._h, float(fraction)))

    def set_prune_interval(self, frames):
        """LatticeFasterDecoderConfig.prune_interval for streams: PruneActiveTokens (arena compaction) of a stream every `frames`
        decoded frames, between ticks; FinalizeDecoding then finds all but the last frames pruned.  0 = never (default)."""
        check(lib().kamd_stream_batch_set_prune_interval(self._h, int(frames)))

    def num_compactions(self):
        return int(lib().kamd_stream_batch_num_compactions(self._h))

    def frame_tracebacks(self, streams):
        return decoder.frame_tracebacks(self.dec._dec, streams)

    def start(self, streams, states=None):
        """states: one adaptation state per stream (the speaker's, after LimitFrames) or None = fresh"""
        s = np.ascontiguousarray(streams, np.int32)
        if states is None:
            check(lib().kamd_stream_batch_start(self._h, abi.iptr(s), s.size))
        else:
            st = np.ascontiguousarray(states, np.float64).reshape(s.size, -1)
            check(lib().kamd_stream_batch_start_adapted(self._h, abi.iptr(s), s.size, st.ctypes.data_as(C.POINTER(C.c_double))))

    def adaptation_state(self, stream, max_remembered_frames=1000.0):
        """GetAdaptationState + LimitFrames after the stream's utterance: what the speaker's next utterance starts from"""
        st = np.zeros(self.extractor.info.state_size(), np.float64)
        dp = C.POINTER(C.c_double)
        check(lib().kamd_stream_batch_get_adaptation_state(self._h, int(stream), st.ctypes.data_as(dp)))
        check(lib().kamd_ivector_state_limit_frames(self.extractor._h, st.ctypes.data_as(dp), max_remembered_frames))
        return st

    def ivector_slots(self, stream):
        """(first slot number, [slots x dim]): slot j covers first-layer times [j * frames_per_chunk, +frames_per_chunk)"""
        cap = 4096
        out = np.zeros((cap, self.extractor.dim()), np.float32)
        first, cnt = C.c_int32(), C.c_int32()
        check(lib().kamd_stream_batch_get_ivector_slots(self._h, int(stream), abi.fptr(out), cap, C.byref(first), C.byref(cnt)))
        return first.value, out[:cnt.value].copy()

    def num_frames_ready(self, stream):
        return lib().kamd_stream_batch_num_frames_ready(self._h, int(stream))

    def accept(self, stream, waveform, input_finished=False):
        w = np.ascontiguousarray(waveform, np.float32)
        check(lib().kamd_stream_batch_accept(self._h, int(stream), abi.fptr(w), w.size, int(input_finished)))

    def accept_many(self, streams, chunks, input_finished=None):
        """AcceptWaveform for several streams with ONE upload: chunks[i] is appended to streams[i]"""
        st = np.ascontiguousarray(streams, np.int32)
        chunks = [np.ascontiguousarray(c, np.float32).reshape(-1) for c in chunks]
        if len(chunks) != st.size:
            raise KamdError("accept_many: one chunk per stream")
        off = np.zeros(st.size + 1, np.int64)
        off[1:] = np.cumsum([c.size for c in chunks])
        flat = np.ascontiguousarray(np.concatenate(chunks)) if off[-1] else np.zeros(1, np.float32)
        fin = None if input_finished is None else np.ascontiguousarray([int(bool(f)) for f in input_finished], np.int32)
        check(lib().kamd_stream_batch_accept_many(self._h, abi.iptr(st), st.size, abi.fptr(flat), abi.iptr(off, C.c_int64),
                                                  None if fin is None else abi.iptr(fin)))

    def advance(self, streams):
        """One tick for these streams; returns NumFramesDecoded of each."""
        s = np.ascontiguousarray(streams, np.int32)
        out = np.zeros(s.size, np.int32)
        check(lib().kamd_stream_batch_advance(self._h, abi.iptr(s), s.size, abi.iptr(out)))
        return out

    def status(self, streams):
        """0 = fine; otherwise the capacity flags that took the stream out of service (restart it with start())."""
        s = np.ascontiguousarray(streams, np.int32)
        out = np.zeros(s.size, np.int32)
        check(lib().kamd_stream_batch_get_status(self._h, abi.iptr(s), s.size, abi.iptr(out)))
        return out

    def partial_best_path(self, stream, use_final_probs=False):
        return decoder.partial_best_path(self.dec._dec, int(stream), use_final_probs)

    def partial_best_paths(self, streams, use_final_probs=False, incremental=False):
        """partial results of these streams in one launch; incremental: only the frames whose best-path token changed since
        the last call are walked (kamd_decoder_partial_best_paths_incremental)"""
        return decoder.partial_best_paths(self.dec._dec, streams, use_final_probs, incremental)

    def endpoint_detected(self, config, streams, tid2phone, silence_phones, frame_shift_in_seconds=None):
        """EndpointDetected for these streams in one launch -> (flags, trailing silence frames)"""
        if frame_shift_in_seconds is None:
            frame_shift_in_seconds = self.feat.opts.frame.frame_shift_ms * 1e-3 * lib().kamd_nnet_frame_subsampling_factor(self.nnet._h)
        _set_silence(self.dec, tid2phone, silence_phones)
        return decoder.endpoint_detected(self.dec._dec, config.to_c(), streams, frame_shift_in_seconds)

    def finalize(self, streams):
        s = np.ascontiguousarray(streams, np.int32)
        check(lib().kamd_decoder_finalize(self.dec._dec, abi.iptr(s), s.size, None))
        err = np.zeros(s.size, np.int32)
        check(lib().kamd_decoder_sync_lanes(self.dec._dec, abi.iptr(s), s.size, abi.iptr(err)))
        if err.any():
            raise KamdError("stream(s) %s exceeded their lane's capacity (flags %s)" % (s[err != 0].tolist(), err[err != 0].tolist()))

    def raw_lattice(self, stream):
        return decoder.get_raw_lattice(self.dec._dec, int(stream))

    def best_path(self, stream):
        return decoder.best_path(self.dec._dec, int(stream))
