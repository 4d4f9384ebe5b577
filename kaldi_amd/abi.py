"""ctypes mirror of include/kaldi_amd.h (POD structs + defaults).

Pure interface definitions: no compute, no library loading.  Used by the product
wrappers (kaldi_amd/_lib.py) and, for the struct layouts only, by the test oracle.
Defaults follow the reference option structs cited in the header.
"""
import ctypes as C

import numpy as np

KAMD_WIN = {"hanning": 0, "hamming": 1, "povey": 2, "rectangular": 3, "blackman": 4}
KAMD_MAX_OFFSETS = 8
INT32_MAX = 2147483647


class FrameOpts(C.Structure):
    """feat/feature-window.h:40-66 FrameExtractionOptions."""
    _fields_ = [("samp_freq", C.c_float), ("frame_shift_ms", C.c_float),
                ("frame_length_ms", C.c_float), ("dither", C.c_float),
                ("preemph_coeff", C.c_float), ("remove_dc_offset", C.c_int32),
                ("window_type", C.c_int32), ("round_to_power_of_two", C.c_int32),
                ("blackman_coeff", C.c_float), ("snip_edges", C.c_int32)]


class MelOpts(C.Structure):
    """feat/mel-computations.h:43-57 MelBanksOptions."""
    _fields_ = [("num_bins", C.c_int32), ("low_freq", C.c_float), ("high_freq", C.c_float),
                ("vtln_low", C.c_float), ("vtln_high", C.c_float), ("htk_mode", C.c_int32)]


class MfccOpts(C.Structure):
    """feat/feature-mfcc.h:38-56 MfccOptions."""
    _fields_ = [("frame", FrameOpts), ("mel", MelOpts), ("num_ceps", C.c_int32),
                ("use_energy", C.c_int32), ("energy_floor", C.c_float),
                ("raw_energy", C.c_int32), ("cepstral_lifter", C.c_float),
                ("htk_compat", C.c_int32)]


class FbankOpts(C.Structure):
    """feat/feature-fbank.h FbankOptions."""
    _fields_ = [("frame", FrameOpts), ("mel", MelOpts), ("use_energy", C.c_int32),
                ("energy_floor", C.c_float), ("raw_energy", C.c_int32),
                ("htk_compat", C.c_int32), ("use_log_fbank", C.c_int32),
                ("use_power", C.c_int32)]


def frame_opts_default():
    # NOTE dither: the reference default is 1.0 (random); every deterministic path
    # here needs dither=0 and the constructors below set that explicitly.
    return FrameOpts(16000.0, 10.0, 25.0, 0.0, 0.97, 1, KAMD_WIN["povey"], 1, 0.42, 1)


def mfcc_opts_default():
    o = MfccOpts()
    o.frame = frame_opts_default()
    o.mel = MelOpts(23, 20.0, 0.0, 100.0, -500.0, 0)
    o.num_ceps, o.use_energy, o.energy_floor = 13, 1, 0.0
    o.raw_energy, o.cepstral_lifter, o.htk_compat = 1, 22.0, 0
    return o


def mfcc_opts_hires():
    """egs/mini_librispeech/s5/conf/mfcc_hires.conf (40-dim, no energy)."""
    o = mfcc_opts_default()
    o.use_energy = 0
    o.mel.num_bins = 40
    o.num_ceps = 40
    o.mel.low_freq = 20.0
    o.mel.high_freq = -400.0
    return o


def fbank_opts_default():
    o = FbankOpts()
    o.frame = frame_opts_default()
    o.mel = MelOpts(23, 20.0, 0.0, 100.0, -500.0, 0)
    o.use_energy, o.energy_floor, o.raw_energy = 0, 0.0, 1
    o.htk_compat, o.use_log_fbank, o.use_power = 0, 1, 1
    return o


class LayerDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("n_offsets", C.c_int32),
                ("offsets", C.c_int32 * KAMD_MAX_OFFSETS), ("input_layer", C.c_int32),
                ("ivector_dim", C.c_int32), ("bypass_layer", C.c_int32),
                ("bypass_scale", C.c_float), ("relu", C.c_int32), ("log_softmax", C.c_int32),
                ("W", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("bn_scale", C.POINTER(C.c_float)), ("bn_offset", C.POINTER(C.c_float)),
                ("post_offset", C.POINTER(C.c_float)), ("post_scale", C.c_float),
                ("multi_input", C.c_int32), ("slice_layer", C.c_int32 * KAMD_MAX_OFFSETS),
                ("slice_dim", C.c_int32 * KAMD_MAX_OFFSETS)]


class Arc(C.Structure):
    """== fst::StdArc layout."""
    _fields_ = [("ilabel", C.c_int32), ("olabel", C.c_int32), ("weight", C.c_float),
                ("nextstate", C.c_int32)]


ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"),
                      ("nextstate", "<i4")])


class DecoderConfig(C.Structure):
    """decoder/lattice-faster-decoder.h:38-64 LatticeFasterDecoderConfig."""
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("lattice_beam", C.c_float), ("prune_interval", C.c_int32),
                ("beam_delta", C.c_float), ("hash_ratio", C.c_float),
                ("prune_scale", C.c_float)]


def decoder_config_default():
    return DecoderConfig(16.0, INT32_MAX, 200, 10.0, 25, 0.5, 2.0, 0.1)


def decoder_config_recipe():
    """steps/nnet3/decode.sh:14-20: beam 15, max-active 7000, min-active 200, lattice-beam 8."""
    c = decoder_config_default()
    c.beam, c.max_active, c.min_active, c.lattice_beam = 15.0, 7000, 200, 8.0
    return c


class DecoderSizes(C.Structure):
    _fields_ = [("max_lanes", C.c_int32), ("hash_capacity", C.c_int32),
                ("arena_tokens", C.c_int64), ("arena_links", C.c_int64),
                ("max_frames", C.c_int32)]


class DecodeTask(C.Structure):
    _fields_ = [("lane", C.c_int32), ("n_frames", C.c_int32), ("d_loglikes", C.c_void_p),
                ("ld", C.c_int32), ("reserved", C.c_int32)]


class QueueTask(C.Structure):
    """kamd_queue_task: one utterance of a work-queue launch."""
    _fields_ = [("d_loglikes", C.c_void_p), ("ld", C.c_int32), ("n_frames", C.c_int32),
                ("utt", C.c_int32), ("reserved", C.c_int32)]


class QueueResult(C.Structure):
    """kamd_queue_result: the record a lane publishes when an utterance is done."""
    _fields_ = [("status", C.c_int32), ("error", C.c_int32), ("lane", C.c_int32), ("n_frames", C.c_int32),
                ("n_tok", C.c_int32), ("n_link", C.c_int32), ("n_last", C.c_int32), ("n_preselected", C.c_int32),
                ("final_relative_cost", C.c_float), ("final_best_cost", C.c_float),
                ("blob_off", C.c_int64), ("blob_bytes", C.c_int64), ("counters", C.c_int64 * 8),
                ("phase_cycles", C.c_uint64 * 16)]


class LatticeSize(C.Structure):
    _fields_ = [("num_states", C.c_int32), ("num_arcs", C.c_int32),
                ("num_frames", C.c_int32), ("start", C.c_int32)]


class IvectorDesc(C.Structure):
    """kamd_ivector_desc (include/kaldi_amd.h)."""
    _fields_ = [("feat_dim", C.c_int32), ("splice_left", C.c_int32), ("splice_right", C.c_int32),
                ("lda_rows", C.c_int32), ("lda_cols", C.c_int32), ("lda", C.POINTER(C.c_float)),
                ("global_cmvn_stats", C.POINTER(C.c_double)),
                ("cmn_window", C.c_int32), ("speaker_frames", C.c_int32), ("global_frames", C.c_int32),
                ("normalize_mean", C.c_int32), ("normalize_variance", C.c_int32),
                ("num_gauss", C.c_int32), ("ubm_gconsts", C.POINTER(C.c_float)),
                ("ubm_means_invvars", C.POINTER(C.c_float)), ("ubm_inv_vars", C.POINTER(C.c_float)),
                ("ivector_dim", C.c_int32), ("M", C.POINTER(C.c_double)), ("sigma_inv", C.POINTER(C.c_double)),
                ("prior_offset", C.c_double),
                ("ivector_period", C.c_int32), ("num_gselect", C.c_int32), ("num_cg_iters", C.c_int32),
                ("min_post", C.c_float), ("posterior_scale", C.c_float), ("max_count", C.c_float)]


class EndpointRule(C.Structure):
    """online2/online-endpoint.h:113-143 OnlineEndpointRule."""
    _fields_ = [("must_contain_nonsilence", C.c_int32), ("min_trailing_silence", C.c_float),
                ("max_relative_cost", C.c_float), ("min_utterance_length", C.c_float)]


class EndpointConfig(C.Structure):
    """online2/online-endpoint.h:145-170 OnlineEndpointConfig (rule1..rule5); the decoder gets the silence
    phones by kamd_decoder_set_silence_phones."""
    _fields_ = [("rule", EndpointRule * 5)]


class DeterminizeOpts(C.Structure):
    _fields_ = [("delta", C.c_float), ("max_mem", C.c_int32), ("phone_determinize", C.c_int32),
                ("word_determinize", C.c_int32), ("max_loop", C.c_int32), ("retry_cutoff", C.c_float)]


class BatchOpts(C.Structure):
    """kamd_batch_opts."""
    _fields_ = [("resident_lanes", C.c_int32), ("host_threads", C.c_int32), ("determinize", C.c_int32),
                ("keep_raw_lattices", C.c_int32), ("nnet_pass_frames", C.c_int64), ("lattice_pool_bytes", C.c_int64),
                ("lattice_beam", C.c_float), ("det", DeterminizeOpts), ("first_pass_frames", C.c_int64)]


class BatchStats(C.Structure):
    """kamd_batch_stats."""
    _fields_ = [("feat_ms", C.c_float), ("nnet_ms", C.c_float), ("decode_ms", C.c_float), ("host_tail_ms", C.c_float),
                ("first_result_ms", C.c_float), ("total_ms", C.c_float), ("nnet_flops", C.c_double),
                ("host_thread_ms_sum", C.c_double), ("lanes", C.c_int32), ("nnet_passes", C.c_int32),
                ("n_failed", C.c_int32), ("long_utterances", C.c_int32), ("upload_ms", C.c_float),
                ("first_pass_start_ms", C.c_float), ("upload_wait_ms", C.c_float), ("upload_passes", C.c_int32),
                ("ivector_ms", C.c_float), ("n_retried", C.c_int32), ("n_internal_events", C.c_int32)]


CLAT_ARC_DTYPE = np.dtype([("src", "<i4"), ("dst", "<i4"), ("label", "<i4"), ("graph_cost", "<f4"),
                           ("acoustic_cost", "<f4"), ("str_begin", "<i4"), ("str_len", "<i4")])
LAT_ARC_DTYPE = np.dtype([("src", "<i4"), ("dst", "<i4"), ("ilabel", "<i4"),
                          ("olabel", "<i4"), ("graph_cost", "<f4"),
                          ("acoustic_cost", "<f4")])


def fptr(a):
    """float* of a C-contiguous float32 numpy array (or NULL)."""
    if a is None:
        return C.POINTER(C.c_float)()
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a, t=C.c_int32):
    if a is None:
        return C.POINTER(t)()
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(t))


class InferenceTask(C.Structure):
    """kamd_inference_task: one NnetInferenceTask of a minibatch (kamd_nnet_forward_inference_tasks_device)"""
    _fields_ = [("in_row", C.c_int64), ("in_len", C.c_int32), ("first_output_t", C.c_int32), ("num_output_frames", C.c_int32),
                ("iv_row", C.c_int32)]
