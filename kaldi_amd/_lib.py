"""Loader for the C-ABI shared library (kaldi_amd/lib/libkaldi_amd.so).

The HIP library IS the product: if it is missing or cannot be loaded this module raises
-- there is no CPU fallback anywhere in the package.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KAMD_LIB") or os.path.join(_HERE, "lib", "libkaldi_amd.so")
_LIB = None


class KamdError(RuntimeError):
    """Raised for a non-zero C-ABI status (mirrors KALDI_ERR -> KaldiFatalError,
    base/kaldi-error.h:89-140)."""


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise KamdError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    fp, ip, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    vp = C.c_void_p

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, args

    sig("kamd_last_error", C.c_char_p, [])
    sig("kamd_version", C.c_char_p, [])
    sig("kamd_device_count", C.c_int, [])
    sig("kamd_set_device", C.c_int, [C.c_int])
    sig("kamd_device_num_cus", C.c_int, [])
    sig("kamd_malloc", vp, [C.c_size_t])
    sig("kamd_free", C.c_int, [vp])
    sig("kamd_memcpy_h2d", C.c_int, [vp, vp, C.c_size_t])
    sig("kamd_memcpy_d2h", C.c_int, [vp, vp, C.c_size_t])
    sig("kamd_device_synchronize", C.c_int, [])
    sig("kamd_device_mem_info", C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)])
    sig("kamd_mfcc_opts_default", None, [C.POINTER(abi.MfccOpts)])
    sig("kamd_fbank_opts_default", None, [C.POINTER(abi.FbankOpts)])
    sig("kamd_mfcc_create", vp, [C.POINTER(abi.MfccOpts), C.c_float])
    sig("kamd_fbank_create", vp, [C.POINTER(abi.FbankOpts), C.c_float])
    sig("kamd_feat_destroy", None, [vp])
    sig("kamd_feat_dim", C.c_int, [vp])
    sig("kamd_feat_num_frames", C.c_int, [vp, C.c_int64])
    sig("kamd_feat_compute", C.c_int, [vp, fp, C.c_int64, fp, C.c_int])
    sig("kamd_feat_compute_batch_device", C.c_int, [vp, vp, i64p, C.c_int, vp, i64p, C.c_int, vp])
    sig("kamd_feat_num_frames_flush", C.c_int, [vp, C.c_int64, C.c_int])
    sig("kamd_feat_compute_frames_device", C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int, vp])
    sig("kamd_online_feat_create", vp, [vp])
    sig("kamd_online_feat_destroy", None, [vp])
    sig("kamd_online_feat_accept_waveform", C.c_int, [vp, C.c_float, fp, C.c_int64])
    sig("kamd_online_feat_input_finished", C.c_int, [vp])
    sig("kamd_online_feat_num_frames_ready", C.c_int, [vp])
    sig("kamd_online_feat_is_last_frame", C.c_int, [vp, C.c_int])
    sig("kamd_online_feat_get_frames", C.c_int, [vp, C.c_int, C.c_int, fp])
    sig("kamd_online_feat_device_frames", vp, [vp, ip])
    sig("kamd_nnet_num_frames_ready", C.c_int, [vp, C.c_int, C.c_int])
    sig("kamd_nnet_forward_range", C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int])
    sig("kamd_nnet_create", vp, [C.POINTER(abi.LayerDesc), C.c_int, C.c_int, C.c_int])
    sig("kamd_component_create", vp, [C.POINTER(abi.LayerDesc)])
    sig("kamd_component_destroy", None, [vp])
    sig("kamd_component_output_rows", C.c_int, [vp, C.c_int])
    sig("kamd_component_propagate", C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp])
    sig("kamd_model_read", vp, [C.c_char_p, C.c_float, C.c_int])
    sig("kamd_model_destroy", None, [vp])
    sig("kamd_model_info", C.c_int, [vp, ip, ip, ip, ip, ip, ip])
    sig("kamd_model_layers", C.POINTER(abi.LayerDesc), [vp])
    sig("kamd_model_transition_tables", C.c_int, [vp, ip, ip, ip])
    sig("kamd_model_create_nnet", vp, [vp])
    sig("kamd_nnet_destroy", None, [vp])
    sig("kamd_nnet_output_dim", C.c_int, [vp])
    sig("kamd_nnet_left_context", C.c_int, [vp])
    sig("kamd_nnet_right_context", C.c_int, [vp])
    sig("kamd_nnet_num_output_frames", C.c_int, [vp, C.c_int])
    sig("kamd_nnet_frame_subsampling_factor", C.c_int, [vp])
    sig("kamd_nnet_forward_batch_device", C.c_int, [vp, vp, i64p, C.c_int, vp, C.c_int, vp, i64p, C.c_int, vp])
    sig("kamd_nnet_forward", C.c_int, [vp, fp, C.c_int, fp, fp, C.c_int])
    sig("kamd_nnet_last_flops", C.c_double, [vp])
    sig("kamd_graph_create", vp, [C.c_int32, C.c_int32, i64p, vp, fp])
    sig("kamd_graph_read_openfst", vp, [C.c_char_p])
    sig("kamd_openfst_read", C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.POINTER(C.c_int64)), C.POINTER(vp), C.POINTER(C.POINTER(C.c_float))])
    sig("kamd_openfst_write", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int32, C.c_int32, i64p, vp, fp])
    sig("kamd_host_free", None, [vp])
    sig("kamd_lattice_write", C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int32, C.c_int32, fp, vp, C.c_int32])
    sig("kamd_lattice_read", C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float)), C.POINTER(vp),
                                       C.POINTER(C.c_int32)])
    sig("kamd_determinize_opts_default", None, [C.POINTER(abi.DeterminizeOpts)])
    sig("kamd_lattice_determinize_phone_pruned", vp, [C.c_int32, C.c_int32, fp, vp, C.c_int32, ip, C.c_int32, C.c_double,
                                                      C.POINTER(abi.DeterminizeOpts)])
    sig("kamd_compact_lattice_destroy", None, [vp])
    sig("kamd_compact_lattice_sizes", C.c_int, [vp] + [C.POINTER(C.c_int32)] * 5)
    sig("kamd_compact_lattice_get", C.c_int, [vp, fp, ip, ip, vp, ip])
    sig("kamd_compact_lattice_write", C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, vp, C.c_float])
    sig("kamd_stream_batch_create", vp, [vp, vp, vp, C.c_int, C.c_float, C.c_float])
    sig("kamd_stream_batch_destroy", None, [vp])
    sig("kamd_stream_batch_start", C.c_int, [vp, ip, C.c_int])
    sig("kamd_stream_batch_accept", C.c_int, [vp, C.c_int, fp, C.c_int64, C.c_int])
    sig("kamd_stream_batch_accept_many", C.c_int, [vp, ip, C.c_int, fp, i64p, ip])
    sig("kamd_stream_batch_advance", C.c_int, [vp, ip, C.c_int, ip])
    sig("kamd_stream_batch_num_frames_ready", C.c_int, [vp, C.c_int])
    sig("kamd_stream_batch_get_status", C.c_int, [vp, ip, C.c_int, ip])
    sig("kamd_decoder_sync_lanes", C.c_int, [vp, ip, C.c_int, ip])
    sig("kamd_stream_batch_set_ivector_extractor", C.c_int, [vp, vp, C.c_int, C.c_int])
    sig("kamd_stream_batch_start_adapted", C.c_int, [vp, ip, C.c_int, C.POINTER(C.c_double)])
    sig("kamd_stream_batch_get_adaptation_state", C.c_int, [vp, C.c_int, C.POINTER(C.c_double)])
    sig("kamd_stream_batch_get_ivector_slots", C.c_int, [vp, C.c_int, fp, C.c_int, ip, ip])
    sig("kamd_feat_compute_ranges_device", C.c_int, [vp, vp, i64p, i64p, ip, ip, C.c_int, vp, i64p, C.c_int, vp])
    sig("kamd_nnet_forward_slices_device", C.c_int, [vp, vp, i64p, ip, C.c_int, vp, C.c_int, vp, i64p, C.c_int, vp])
    sig("kamd_nnet_forward_slices_slots_device", C.c_int, [vp, vp, i64p, ip, C.c_int, vp, C.c_int, C.c_int, ip, ip, ip, ip, C.c_int, vp,
                                                           i64p, C.c_int, vp])
    sig("kamd_nnet_forward_chunked_device", C.c_int, [vp, vp, i64p, C.c_int, vp, i64p, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                                      i64p, C.c_int, vp])
    sig("kamd_nnet_forward_tasks_device", C.c_int, [vp, vp, i64p, C.c_int, vp, i64p, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                                    i64p, C.c_int, vp])
    sig("kamd_batch_decoder_set_chunk_rule", C.c_int, [vp, C.c_int])
    sig("kamd_pipeline_set_online_ivectors", C.c_int, [vp, fp, i64p, C.c_int, C.c_int, C.c_int])
    sig("kamd_wave_read", C.c_int, [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                    C.POINTER(C.POINTER(C.c_float))])
    sig("kamd_ark_read_matrix", C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int, C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float))])
    sig("kamd_ark_write_matrix", C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int32, C.c_int32, fp])
    sig("kamd_ark_read_int32_vector", C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int, C.POINTER(C.c_int32),
                                                C.POINTER(C.POINTER(C.c_int32))])
    sig("kamd_classify_rxfilename", C.c_int, [C.c_char_p])
    sig("kamd_classify_wxfilename", C.c_int, [C.c_char_p])
    sig("kamd_classify_rspecifier", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, ip])
    sig("kamd_classify_wspecifier", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, ip])
    sig("kamd_rx_materialize", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64), ip])
    sig("kamd_graph_destroy", None, [vp])
    sig("kamd_graph_num_states", C.c_int32, [vp])
    sig("kamd_graph_num_arcs", C.c_int64, [vp])
    sig("kamd_decoder_config_default", None, [C.POINTER(abi.DecoderConfig)])
    sig("kamd_decoder_sizes_default", None, [C.POINTER(abi.DecoderSizes)])
    sig("kamd_decoder_sizes_suggest", C.c_int, [C.POINTER(abi.DecoderConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.POINTER(abi.DecoderSizes)])
    sig("kamd_decoder_create", vp, [vp, C.POINTER(abi.DecoderConfig), C.POINTER(abi.DecoderSizes), ip, C.c_int32])
    sig("kamd_decoder_destroy", None, [vp])
    sig("kamd_decoder_set_options", C.c_int, [vp, C.POINTER(abi.DecoderConfig)])
    sig("kamd_decoder_set_search_mode", C.c_int, [vp, C.c_int])
    sig("kamd_decoder_set_token_preselection", C.c_int, [vp, C.c_int])
    sig("kamd_decoder_lds_layout", C.c_int, [vp, ip, ip])
    sig("kamd_decoder_set_level1_table", C.c_int, [vp, C.c_int32])
    sig("kamd_decoder_lanes_per_cu", C.c_int, [])
    sig("kamd_decoder_reserve", C.c_int, [vp, ip, C.c_int])
    sig("kamd_decoder_init", C.c_int, [vp, ip, C.c_int, vp])
    sig("kamd_decoder_advance", C.c_int, [vp, C.POINTER(abi.DecodeTask), C.c_int, vp])
    sig("kamd_decoder_finalize", C.c_int, [vp, ip, C.c_int, vp])
    sig("kamd_decoder_sync", C.c_int, [vp])
    sig("kamd_decoder_num_frames_decoded", C.c_int, [vp, C.c_int])
    sig("kamd_decoder_final_relative_cost", C.c_float, [vp, C.c_int])
    sig("kamd_decoder_reached_final", C.c_int, [vp, C.c_int])
    sig("kamd_decoder_lattice_size", C.c_int, [vp, C.c_int, C.POINTER(abi.LatticeSize)])
    sig("kamd_decoder_get_raw_lattice", C.c_int, [vp, C.c_int, ip, ip, fp, fp, vp])
    sig("kamd_decoder_best_path", C.c_int, [vp, C.c_int, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp])
    sig("kamd_decoder_partial_best_path", C.c_int, [vp, C.c_int, C.c_int, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp])
    sig("kamd_decoder_partial_best_paths", C.c_int, [vp, ip, C.c_int, C.c_int, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp])
    sig("kamd_decoder_partial_best_paths_incremental", C.c_int, [vp, ip, C.c_int, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp])
    sig("kamd_endpoint_config_default", None, [C.POINTER(abi.EndpointConfig)])
    sig("kamd_endpoint_detected", C.c_int, [C.POINTER(abi.EndpointConfig), C.c_int, C.c_int, C.c_float, C.c_float])
    sig("kamd_decoder_set_silence_phones", C.c_int, [vp, ip, C.c_int, ip, C.c_int])
    sig("kamd_decoder_trailing_silence_frames", C.c_int, [vp, ip, C.c_int, ip])
    sig("kamd_decoder_endpoint_detected", C.c_int, [vp, C.POINTER(abi.EndpointConfig), ip, C.c_int, C.c_float, ip, ip])
    sig("kamd_decoder_get_trace", C.c_int, [vp, C.c_int, ip, fp, fp, C.c_int])
    sig("kamd_decoder_get_counters", C.c_int, [vp, C.c_int, i64p])
    sig("kamd_decoder_get_phase_cycles", C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64)])
    sig("kamd_decoder_last_advance_ms", C.c_float, [vp])
    sig("kamd_decoder_queue_configure", C.c_int, [vp, C.c_int64])
    sig("kamd_decoder_queue_launch", C.c_int, [vp, C.POINTER(abi.QueueTask), C.c_int, C.c_int, vp])
    sig("kamd_decoder_queue_poll", C.c_int, [vp, ip, C.c_int])
    sig("kamd_decoder_queue_result", C.c_int, [vp, C.c_int32, C.POINTER(abi.QueueResult)])
    sig("kamd_decoder_queue_fetch_lattice", C.c_int, [vp, C.c_int32, vp, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(fp),
                                                      C.POINTER(fp), C.POINTER(vp)])
    sig("kamd_decoder_queue_wait", C.c_int, [vp, fp, ip])
    sig("kamd_compact_lattice_scale_graph", C.c_int, [vp, C.c_float])
    sig("kamd_const_arpa_build", vp, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p])
    sig("kamd_const_arpa_read", vp, [C.c_char_p])
    sig("kamd_const_arpa_write", C.c_int, [vp, C.c_char_p])
    sig("kamd_const_arpa_destroy", None, [vp])
    sig("kamd_const_arpa_info", C.c_int, [vp, ip, ip, ip, ip, ip, i64p])
    sig("kamd_const_arpa_ngram_logprob", C.c_float, [vp, C.c_int32, ip, C.c_int])
    sig("kamd_arpa_parse", C.c_int, [C.c_char_p, C.c_char_p, ip, C.c_int, ip, ip, ip, ip, fp, fp, C.c_int, ip])
    sig("kamd_compact_lattice_lmrescore_const_arpa", vp, [C.c_int32, C.c_int32, fp, ip, ip, vp, C.c_int32, ip, vp, C.c_float])
    sig("kamd_batch_opts_default", None, [C.POINTER(abi.BatchOpts)])
    sig("kamd_batch_decoder_create", vp, [vp, vp, vp, C.POINTER(abi.BatchOpts), ip, C.c_int32])
    sig("kamd_batch_decoder_destroy", None, [vp])
    sig("kamd_batch_decoder_load", C.c_int, [vp, fp, i64p, C.c_int])
    u8p = C.POINTER(C.c_uint8)
    sig("kamd_silence_weighting_create", vp, [u8p, C.c_int, C.c_float, C.c_float, C.c_int])
    sig("kamd_silence_weighting_destroy", None, [vp])
    sig("kamd_silence_weighting_reset", C.c_int, [vp])
    sig("kamd_silence_weighting_compute_traceback", C.c_int, [vp, C.c_int, ip, ip, C.c_int])
    sig("kamd_silence_weighting_get_delta_weights", C.c_int, [vp, C.c_int, ip])
    sig("kamd_silence_weighting_pop_until", C.c_int, [vp, C.c_int, ip, fp, C.c_int, ip])
    sig("kamd_silence_weighting_num_pending", C.c_int, [vp])
    sig("kamd_decoder_frame_tracebacks", C.c_int, [vp, ip, C.c_int, ip, ip, C.c_int, ip])
    sig("kamd_stream_batch_set_compaction", C.c_int, [vp, C.c_float])
    sig("kamd_stream_batch_set_prune_interval", C.c_int, [vp, C.c_int])
    sig("kamd_stream_batch_num_compactions", C.c_int64, [vp])
    sig("kamd_decoder_compact", C.c_int, [vp, ip, C.c_int, vp])
    sig("kamd_decoder_lane_usage", C.c_int, [vp, C.c_int, ip, ip, ip, ip])
    sig("kamd_decoder_frame_tracebacks_incremental", C.c_int, [vp, ip, C.c_int, ip, ip, C.c_int, ip, ip])
    sig("kamd_stream_batch_set_silence_weighting", C.c_int, [vp, u8p, C.c_int, C.c_float, C.c_float])
    sig("kamd_ivector_stream_update_weighted_device", C.c_int, [vp, vp, C.c_int, C.c_int64, i64p, ip, ip, ip, ip, ip, ip, fp, C.c_int, vp, vp, vp])
    sig("kamd_compact_lattice_scale", C.c_int, [vp, C.c_float, C.c_float])
    sig("kamd_compact_lattice_copy", vp, [vp])
    sig("kamd_batch_decoder_load_features", C.c_int, [vp, fp, i64p, C.c_int, fp, C.c_int, C.c_int])
    sig("kamd_nnet_input_dim", C.c_int, [vp])
    sig("kamd_nnet_ivector_dim", C.c_int, [vp])
    sig("kamd_batch_decoder_run", C.c_int, [vp, C.POINTER(abi.BatchStats)])
    sig("kamd_batch_decoder_set_long_decoder", C.c_int, [vp, vp, C.c_int])
    sig("kamd_decoder_live_lattice_size", C.c_int, [vp, C.c_int, C.c_int, C.POINTER(abi.LatticeSize)])
    sig("kamd_decoder_get_live_raw_lattice", C.c_int, [vp, C.c_int, C.c_int, ip, ip, fp, fp, vp])
    sig("kamd_lattice_prune", C.c_int, [C.c_int32, C.c_int32, fp, vp, C.c_int32, C.c_float, ip, C.POINTER(C.c_uint8), ip, ip])
    sig("kamd_ivector_workspace_create", vp, [])
    sig("kamd_ivector_workspace_destroy", None, [vp])
    sig("kamd_ivector_extractor_bind_workspace", C.c_int, [vp, vp])
    sig("kamd_batch_decoder_load_host", C.c_int, [vp, fp, i64p, C.c_int])
    sig("kamd_batch_decoder_unload_host", C.c_int, [vp])
    sig("kamd_batch_decoder_set_ivector_extractor", C.c_int, [vp, vp, C.c_int])
    sig("kamd_batch_decoder_set_loglike_override", C.c_int, [vp, vp])
    sig("kamd_batch_decoder_output_frames", C.c_int64, [vp, ip, C.c_int])
    sig("kamd_synth_planted_loglikes_device", C.c_int, [vp, C.c_int64, C.c_int, C.c_int, vp, C.c_float, C.c_float, C.c_uint64, vp])
    sig("kamd_batch_decoder_get_output", C.c_int, [vp, C.c_int, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp, C.POINTER(abi.QueueResult)])
    sig("kamd_batch_decoder_get_raw_lattice", C.c_int, [vp, C.c_int, ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(fp),
                                                        C.POINTER(fp), C.POINTER(vp)])
    sig("kamd_batch_decoder_get_compact_lattice", vp, [vp, C.c_int])
    sig("kamd_batch_decoder_get_loglikes", C.c_int, [vp, C.c_int, fp, C.c_int, ip, ip])
    sig("kamd_lattice_best_path", C.c_int, [C.c_int32, C.c_int32, fp, vp, C.c_int32, ip, C.c_int, ip, ip, C.c_int, ip, fp, fp])
    sig("kamd_pipeline_create", vp, [vp, vp, vp])
    sig("kamd_pipeline_destroy", None, [vp])
    sig("kamd_pipeline_load_batch", C.c_int, [vp, fp, i64p, C.c_int])
    sig("kamd_am_gmm_create", vp, [C.c_int32, C.c_int32, ip, fp, fp, fp])
    sig("kamd_am_gmm_destroy", None, [vp])
    sig("kamd_am_gmm_num_pdfs", C.c_int, [vp])
    sig("kamd_am_gmm_dim", C.c_int, [vp])
    sig("kamd_am_gmm_loglikes_device", C.c_int, [vp, vp, C.c_int, C.c_int64, C.c_float, vp, vp])
    sig("kamd_am_gmm_loglikes", C.c_int, [vp, fp, C.c_int, C.c_int, C.c_float, fp])
    sig("kamd_feat_splice_transform_device", C.c_int, [vp, C.c_int, vp, C.c_int, i64p, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, ip,
                                                       C.c_int, C.c_int, vp])
    sig("kamd_feat_add_deltas_device", C.c_int, [vp, C.c_int, vp, C.c_int, i64p, C.c_int, C.c_int, C.c_int, C.c_int, vp])
    sig("kamd_cmvn_acc_stats_device", C.c_int, [vp, i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), vp])
    sig("kamd_cmvn_acc_stats_weighted_device", C.c_int, [vp, i64p, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_double), vp])
    sig("kamd_cmvn_apply_device", C.c_int, [vp, i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, vp])
    sig("kamd_cmvn_apply_reverse_device", C.c_int, [vp, i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, vp])
    sig("kamd_ivector_extractor_create", vp, [C.POINTER(abi.IvectorDesc)])
    sig("kamd_ivector_info_read", vp, [C.c_char_p])
    sig("kamd_ivector_info_destroy", None, [vp])
    sig("kamd_ivector_info_desc", C.POINTER(abi.IvectorDesc), [vp])
    sig("kamd_ivector_info_create_extractor", vp, [vp])
    sig("kamd_ivector_extractor_destroy", None, [vp])
    sig("kamd_ivector_dim", C.c_int, [vp])
    sig("kamd_ivector_period", C.c_int, [vp])
    sig("kamd_ivector_num_ivectors", C.c_int, [vp, C.c_int])
    sig("kamd_ivector_extract_online_device", C.c_int, [vp, vp, i64p, C.c_int, C.c_int, vp, i64p, vp])
    sig("kamd_nnet_forward_inference_tasks_device", C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, vp])
    sig("kamd_ivector_online_reserve_steps", C.c_int, [vp, C.c_int64])
    sig("kamd_ivector_online_stats_device", C.c_int, [vp, vp, i64p, C.c_int, C.c_int, i64p, vp])
    sig("kamd_ivector_online_solve_device", C.c_int, [vp, i64p, C.c_int, vp, i64p, vp])
    sig("kamd_ivector_extract_online_adapt", C.c_int, [vp, fp, C.c_int, fp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)])
    sig("kamd_ivector_stream_record_size", C.c_int, [vp])
    sig("kamd_ivector_stream_record_init", C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)])
    sig("kamd_ivector_stream_update_device", C.c_int, [vp, vp, C.c_int, C.c_int64, i64p, ip, ip, ip, ip, C.c_int, vp, vp, vp])
    sig("kamd_ivector_state_size", C.c_int, [vp])
    sig("kamd_ivector_extract_online_adapt_device", C.c_int, [vp, vp, i64p, C.c_int, C.c_int, vp, i64p, C.POINTER(C.c_double),
                                                              C.POINTER(C.c_double), vp])
    sig("kamd_ivector_state_limit_frames", C.c_int, [vp, C.POINTER(C.c_double), C.c_float])
    sig("kamd_ivector_extract_online", C.c_int, [vp, fp, C.c_int, fp, C.c_int])
    sig("kamd_ivector_last_posteriors", C.c_int, [vp, ip, fp, C.c_int64])
    sig("kamd_pipeline_set_ivector_extractor", C.c_int, [vp, vp, C.c_int])
    sig("kamd_pipeline_set_overlap", C.c_int, [vp, ip, C.c_int])
    sig("kamd_decoder_last_advance_launches", C.c_int, [vp])
    sig("kamd_pipeline_load_features", C.c_int, [vp, fp, i64p, C.c_int, C.c_int])
    sig("kamd_pipeline_set_ivectors", C.c_int, [vp, fp, C.c_int])
    sig("kamd_pipeline_run", C.c_int, [vp, fp])
    sig("kamd_pipeline_get_loglikes", C.c_int, [vp, C.c_int, fp, C.c_int, ip, ip])
    sig("kamd_pipeline_get_features", C.c_int, [vp, C.c_int, fp, C.c_int, ip, ip])
    _LIB = L
    return L


EXPORTS = """kamd_malloc kamd_free kamd_memcpy_h2d kamd_memcpy_d2h kamd_device_synchronize kamd_device_mem_info kamd_last_error kamd_version kamd_device_count kamd_set_device kamd_mfcc_opts_default
kamd_fbank_opts_default kamd_mfcc_create kamd_fbank_create kamd_feat_destroy kamd_feat_dim
kamd_feat_num_frames kamd_feat_compute kamd_feat_compute_batch_device kamd_feat_num_frames_flush kamd_feat_compute_frames_device kamd_online_feat_create kamd_online_feat_destroy kamd_online_feat_accept_waveform kamd_online_feat_input_finished kamd_online_feat_num_frames_ready kamd_online_feat_is_last_frame kamd_online_feat_get_frames kamd_online_feat_device_frames kamd_nnet_num_frames_ready kamd_nnet_forward_range kamd_nnet_create kamd_component_create kamd_component_destroy kamd_component_output_rows kamd_component_propagate kamd_model_read kamd_model_destroy kamd_model_info kamd_model_layers kamd_model_transition_tables kamd_model_create_nnet
kamd_nnet_destroy kamd_nnet_output_dim kamd_nnet_left_context kamd_nnet_right_context
kamd_nnet_num_output_frames kamd_nnet_frame_subsampling_factor kamd_nnet_forward_batch_device kamd_nnet_forward kamd_nnet_last_flops
kamd_graph_create kamd_graph_destroy kamd_graph_num_states kamd_graph_num_arcs
kamd_graph_read_openfst kamd_openfst_read kamd_openfst_write kamd_host_free kamd_lattice_write kamd_lattice_read
kamd_stream_batch_create kamd_stream_batch_destroy kamd_stream_batch_start kamd_stream_batch_accept kamd_stream_batch_accept_many kamd_stream_batch_advance kamd_stream_batch_num_frames_ready kamd_stream_batch_set_ivector_extractor kamd_stream_batch_start_adapted kamd_stream_batch_get_adaptation_state kamd_stream_batch_get_ivector_slots kamd_feat_compute_ranges_device kamd_nnet_forward_slices_device kamd_nnet_forward_slices_slots_device kamd_nnet_forward_chunked_device kamd_nnet_forward_tasks_device kamd_batch_decoder_set_chunk_rule
kamd_wave_read kamd_ark_read_matrix kamd_ark_write_matrix kamd_ark_read_int32_vector
kamd_classify_rxfilename kamd_classify_wxfilename kamd_classify_rspecifier kamd_classify_wspecifier kamd_rx_materialize kamd_pipeline_load_features kamd_pipeline_set_overlap kamd_decoder_last_advance_launches
kamd_am_gmm_create kamd_am_gmm_destroy kamd_am_gmm_num_pdfs kamd_am_gmm_dim kamd_am_gmm_loglikes_device kamd_am_gmm_loglikes kamd_feat_splice_transform_device kamd_feat_add_deltas_device kamd_cmvn_acc_stats_device kamd_cmvn_acc_stats_weighted_device kamd_cmvn_apply_device kamd_cmvn_apply_reverse_device kamd_ivector_extractor_create kamd_ivector_info_read kamd_ivector_info_destroy kamd_ivector_info_desc kamd_ivector_info_create_extractor kamd_ivector_extractor_destroy kamd_ivector_dim kamd_ivector_period kamd_ivector_num_ivectors kamd_ivector_extract_online_device kamd_nnet_forward_inference_tasks_device kamd_ivector_online_reserve_steps kamd_ivector_online_stats_device kamd_ivector_online_solve_device kamd_ivector_extract_online kamd_ivector_last_posteriors kamd_pipeline_set_ivector_extractor kamd_ivector_state_size kamd_ivector_extract_online_adapt_device kamd_ivector_state_limit_frames kamd_ivector_extract_online_adapt kamd_ivector_stream_record_size kamd_ivector_stream_record_init kamd_ivector_stream_update_device
kamd_determinize_opts_default kamd_lattice_determinize_phone_pruned kamd_compact_lattice_destroy kamd_compact_lattice_sizes kamd_compact_lattice_get kamd_compact_lattice_write
kamd_decoder_config_default kamd_decoder_sizes_default kamd_decoder_sizes_suggest kamd_decoder_create kamd_decoder_destroy
kamd_decoder_set_options kamd_decoder_reserve kamd_decoder_init kamd_decoder_advance kamd_decoder_finalize
kamd_decoder_sync kamd_decoder_num_frames_decoded kamd_decoder_final_relative_cost
kamd_decoder_reached_final kamd_decoder_lattice_size kamd_decoder_get_raw_lattice
kamd_decoder_best_path kamd_decoder_partial_best_path kamd_decoder_get_trace kamd_decoder_get_counters
kamd_decoder_partial_best_paths kamd_decoder_partial_best_paths_incremental kamd_endpoint_config_default kamd_endpoint_detected kamd_decoder_set_silence_phones kamd_decoder_trailing_silence_frames kamd_decoder_endpoint_detected
kamd_decoder_get_phase_cycles kamd_decoder_last_advance_ms kamd_pipeline_create kamd_pipeline_destroy kamd_pipeline_load_batch kamd_pipeline_set_ivectors kamd_pipeline_set_online_ivectors
kamd_pipeline_run kamd_pipeline_get_loglikes kamd_pipeline_get_features
kamd_decoder_queue_configure kamd_decoder_queue_launch kamd_decoder_queue_poll kamd_decoder_queue_result kamd_decoder_queue_fetch_lattice kamd_decoder_queue_wait kamd_lattice_best_path
kamd_compact_lattice_scale_graph kamd_const_arpa_build kamd_const_arpa_read kamd_const_arpa_write kamd_const_arpa_destroy kamd_const_arpa_info kamd_const_arpa_ngram_logprob kamd_arpa_parse kamd_compact_lattice_lmrescore_const_arpa kamd_stream_batch_get_status kamd_decoder_sync_lanes kamd_decoder_set_search_mode kamd_decoder_set_token_preselection kamd_decoder_lanes_per_cu kamd_decoder_lds_layout kamd_decoder_set_level1_table kamd_decoder_queue_launch_wide kamd_decoder_max_lanes kamd_decoder_max_frames kamd_device_num_cus kamd_batch_opts_default kamd_batch_decoder_create kamd_batch_decoder_destroy kamd_silence_weighting_create kamd_silence_weighting_destroy kamd_silence_weighting_reset kamd_silence_weighting_compute_traceback kamd_silence_weighting_get_delta_weights kamd_silence_weighting_pop_until kamd_silence_weighting_num_pending kamd_stream_batch_set_compaction kamd_stream_batch_set_prune_interval kamd_stream_batch_num_compactions kamd_decoder_compact kamd_decoder_lane_usage kamd_decoder_frame_tracebacks kamd_decoder_frame_tracebacks_incremental kamd_stream_batch_set_silence_weighting kamd_ivector_stream_update_weighted_device kamd_compact_lattice_scale kamd_compact_lattice_copy kamd_batch_decoder_load kamd_batch_decoder_load_features kamd_nnet_input_dim kamd_nnet_ivector_dim kamd_batch_decoder_run kamd_batch_decoder_get_output kamd_batch_decoder_get_raw_lattice kamd_batch_decoder_get_compact_lattice kamd_batch_decoder_get_loglikes kamd_batch_decoder_set_long_decoder
kamd_batch_decoder_load_host kamd_batch_decoder_unload_host kamd_batch_decoder_set_ivector_extractor kamd_batch_decoder_set_loglike_override kamd_batch_decoder_output_frames
kamd_synth_planted_loglikes_device kamd_ivector_workspace_create kamd_ivector_workspace_destroy kamd_ivector_extractor_bind_workspace
kamd_decoder_live_lattice_size kamd_decoder_get_live_raw_lattice kamd_lattice_prune""".split()


def check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise KamdError(lib().kamd_last_error().decode())
    return rc


def require_gpu():
    n = lib().kamd_device_count()
    if n <= 0:
        raise KamdError("no HIP device visible: the kaldi_amd hot path only runs on an MI355X")
    return n
