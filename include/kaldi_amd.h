/* kaldi_amd.h -- C-ABI of the MI355X-native Kaldi decode hot path.
 *
 * One shared library (kaldi_amd/lib/libkaldi_amd.so) exports exactly the entry
 * points below.  Plain pointers + sizes, POD structs, int status codes, no
 * exceptions, no STL / OpenFst / Kaldi / torch types.  Every entry point cites the
 * reference interface (path:line under the Kaldi tree) it replaces.
 *
 * Conventions
 *   - functions returning int: 0 = ok, negative = error; kamd_last_error() gives
 *     the thread-local message (reference convention: KALDI_ERR throws
 *     KaldiFatalError, base/kaldi-error.h:89-140; a C-ABI must not throw).
 *   - "d_" pointers are DEVICE (HBM) pointers, "h_"/unprefixed are host pointers.
 *   - stream arguments are hipStream_t passed as void* (NULL = default stream).
 *   - costs are positive = bad, log-likelihoods are negated on use, as in
 *     decoder/lattice-faster-decoder.cc.
 */
#ifndef KALDI_AMD_H_
#define KALDI_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAMD_OK 0
#define KAMD_ERR_ARG -1
#define KAMD_ERR_HIP -2
#define KAMD_ERR_CAPACITY -3   /* a device arena / hash overflowed; see message */
#define KAMD_ERR_STATE -4

/* ---------------------------------------------------------------- errors -- */
const char *kamd_last_error(void);
/* library + device probe: returns number of visible HIP devices (>=0) or <0. */
int kamd_device_count(void);
/* compute units of the current device (0 when there is none): the natural number of resident decoder lanes */
int kamd_device_num_cus(void);
int kamd_set_device(int device);
const char *kamd_version(void);
/* HBM buffers for host languages without HIP bindings (cgo / JNI / ctypes callers):
 * thin hipMalloc / hipFree / hipMemcpy wrappers.  Blocking. */
void *kamd_malloc(size_t bytes);
int kamd_free(void *d_ptr);
int kamd_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int kamd_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);
int kamd_device_synchronize(void);
int kamd_device_mem_info(size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo */

/* -------------------------------------------------------------- features -- */
/* feat/feature-window.h:40-66 FrameExtractionOptions (same defaults). */
enum { KAMD_WIN_HANNING = 0, KAMD_WIN_HAMMING = 1, KAMD_WIN_POVEY = 2,
       KAMD_WIN_RECTANGULAR = 3, KAMD_WIN_BLACKMAN = 4 };
typedef struct {
  float samp_freq;        /* 16000 */
  float frame_shift_ms;   /* 10 */
  float frame_length_ms;  /* 25 */
  float dither;           /* reference default 1.0; device path requires 0 */
  float preemph_coeff;    /* 0.97 */
  int32_t remove_dc_offset;       /* 1 */
  int32_t window_type;            /* KAMD_WIN_POVEY */
  int32_t round_to_power_of_two;  /* 1 */
  float blackman_coeff;           /* 0.42 */
  int32_t snip_edges;             /* 1 */
} kamd_frame_opts;

/* feat/mel-computations.h:43-57 MelBanksOptions. */
typedef struct {
  int32_t num_bins;   /* 23 for MFCC default ctor, 40 hires */
  float low_freq;     /* 20 */
  float high_freq;    /* 0: nyquist; <0: offset from nyquist */
  float vtln_low;     /* 100 */
  float vtln_high;    /* -500 */
  int32_t htk_mode;   /* 0 */
} kamd_mel_opts;

/* feat/feature-mfcc.h:38-56 MfccOptions. */
typedef struct {
  kamd_frame_opts frame;
  kamd_mel_opts mel;
  int32_t num_ceps;        /* 13 */
  int32_t use_energy;      /* 1 */
  float energy_floor;      /* 0 */
  int32_t raw_energy;      /* 1 */
  float cepstral_lifter;   /* 22 */
  int32_t htk_compat;      /* 0 */
} kamd_mfcc_opts;

/* feat/feature-fbank.h FbankOptions. */
typedef struct {
  kamd_frame_opts frame;
  kamd_mel_opts mel;
  int32_t use_energy;      /* 0 */
  float energy_floor;      /* 0 */
  int32_t raw_energy;      /* 1 */
  int32_t htk_compat;      /* 0 */
  int32_t use_log_fbank;   /* 1 */
  int32_t use_power;       /* 1 */
} kamd_fbank_opts;

void kamd_mfcc_opts_default(kamd_mfcc_opts *o);
void kamd_fbank_opts_default(kamd_fbank_opts *o);

typedef struct kamd_feat kamd_feat; /* an MfccComputer / FbankComputer on device */

/* replaces MfccComputer::MfccComputer (feat/feature-mfcc.cc:82-115) /
 * FbankComputer ctor; vtln_warp fixes the mel banks (GetMelBanks(vtln_warp)). */
kamd_feat *kamd_mfcc_create(const kamd_mfcc_opts *opts, float vtln_warp);
kamd_feat *kamd_fbank_create(const kamd_fbank_opts *opts, float vtln_warp);
void kamd_feat_destroy(kamd_feat *f);
/* Computer::Dim() (feat/feature-mfcc.h:73). */
int kamd_feat_dim(const kamd_feat *f);
/* NumFrames(num_samples, opts, flush=true) (feat/feature-window.cc:41-87). */
int kamd_feat_num_frames(const kamd_feat *f, int64_t num_samples);
/* OfflineFeatureTpl<F>::ComputeFeatures (feat/feature-common-inl.h:29-83):
 * host wave (int16-range floats) -> host [num_frames x dim] row-major.
 * Returns num_frames (>=0) or <0.  out_rows_cap bounds the output. */
int kamd_feat_compute(kamd_feat *f, const float *wave, int64_t num_samples,
                      float *out, int out_rows_cap);
/* Batched device-resident variant: n_utts waveforms concatenated in d_waves;
 * h_wave_off[n_utts+1] sample offsets (host); output rows of utterance u start at
 * row h_row_off[u] of d_out (leading dimension ld_out floats, >= dim; columns
 * dim..ld_out-1 are zero-filled).  Asynchronous on 'stream'. */
int kamd_feat_compute_batch_device(kamd_feat *f, const float *d_waves,
                                   const int64_t *h_wave_off, int n_utts,
                                   float *d_out, const int64_t *h_row_off,
                                   int ld_out, void *stream);

/* NumFrames with an explicit flush flag, and frames [first_frame, +num_frames) of one
 * device-resident waveform (streaming building blocks). */
int kamd_feat_num_frames_flush(const kamd_feat *f, int64_t num_samples, int flush);
int kamd_feat_compute_frames_device(kamd_feat *f, const float *d_wave, int64_t num_samples,
                                    int first_frame, int num_frames, float *d_out, int ld_out,
                                    void *stream);

/* OnlineGenericBaseFeature<C> (feat/online-feature.h:78, online-feature.cc:63-200;
 * itf/online-feature-itf.h OnlineFeatureInterface): streaming MFCC / fbank.  The
 * computer is borrowed, not owned. */
typedef struct kamd_online_feat kamd_online_feat;
kamd_online_feat *kamd_online_feat_create(kamd_feat *computer);
void kamd_online_feat_destroy(kamd_online_feat *o);
int kamd_online_feat_accept_waveform(kamd_online_feat *o, float sampling_rate, const float *wave, int64_t n);
int kamd_online_feat_input_finished(kamd_online_feat *o);
int kamd_online_feat_num_frames_ready(const kamd_online_feat *o);
int kamd_online_feat_is_last_frame(const kamd_online_feat *o, int frame);
int kamd_online_feat_get_frames(kamd_online_feat *o, int first, int n, float *out /* [n x dim] host */);
/* device view of all frames computed so far ([frames x *ld], zero padded to *ld) */
const float *kamd_online_feat_device_frames(const kamd_online_feat *o, int *ld);

/* ------------------------------------------------------------------ nnet -- */
#define KAMD_MAX_OFFSETS 8
/* One fused layer = TdnnComponent/AffineComponent/LinearComponent/FixedAffine
 * (nnet3/nnet-tdnn-component.cc:181-212, nnet-simple-component.cc:1234,3209,3376)
 * + RectifiedLinear (:957) + test-mode BatchNorm (nnet-normalize-component.cc:
 * 453-464) + the NoOp bypass Sum(Scale(s, prev), this) of tdnnf-layer
 * (steps/libs/nnet3/xconfig/composite_layers.py:201-215) + the final
 * "-log_prior, *acoustic_scale" (nnet-am-decodable-simple.cc:268-271):
 *
 *   a[t] = bn_scale * relu(sum_i W_i x[t + off_i] + W_iv ivec + b) + bn_offset
 *            + bypass_scale * z[t]
 *   if (log_softmax) a[t] -= log(sum_j exp(a[t][j]))
 *   y[t] = (a[t] + post_offset) * post_scale
 *
 * log_softmax = LogSoftmaxComponent (nnet-simple-component.cc:3599,
 * ApplyLogSoftMaxPerRow) of a non-chain "output" node.  FixedScaleComponent,
 * PerElementScale/OffsetComponent and test-mode Dropout scale (:3669,2021,2191,
 * 138-150) are per-element affine maps and go into bn_scale / bn_offset.
 *
 * W is [out_dim x (n_offsets*in_dim + ivector_dim)] row-major, column blocks in
 * time_offsets order (TdnnComponent linear_params_ layout). NULL vectors = absent.
 */
typedef struct {
  int32_t in_dim, out_dim;
  int32_t n_offsets;
  int32_t offsets[KAMD_MAX_OFFSETS];
  int32_t input_layer;    /* producing layer index, -1 = the network input */
  int32_t ivector_dim;    /* >0 only when input_layer == -1 */
  int32_t bypass_layer;   /* -2 = none, -1 = network input, else layer index */
  float bypass_scale;
  int32_t relu;
  int32_t log_softmax;    /* row log-softmax over out_dim before post_offset */
  const float *W;
  const float *bias;       /* [out_dim] or NULL */
  const float *bn_scale;   /* [out_dim] or NULL */
  const float *bn_offset;  /* [out_dim] or NULL */
  const float *post_offset;/* [out_dim] or NULL (= -log_priors) */
  float post_scale;        /* acoustic scale on the last layer, else 1 */
  /* Append over DIFFERENT producers -- descriptors like Append(Offset(tdnn1, -3), tdnn2), which the reference
   * evaluates with kCopyRows / kAddRows over arbitrary sources (nnet3/nnet-compute.cc:309-383).  multi_input != 0:
   * slice i of the layer's input reads slice_layer[i] (-1 = the network input) at time offsets[i] and is slice_dim[i]
   * columns wide; in_dim is the SUM of the slice widths, W's column blocks follow the slices, input_layer is ignored
   * and ivector_dim must be 0.  Zero-initialised descriptors keep the single-producer meaning. */
  int32_t multi_input;
  int32_t slice_layer[KAMD_MAX_OFFSETS];
  int32_t slice_dim[KAMD_MAX_OFFSETS];
} kamd_layer_desc;

typedef struct kamd_nnet kamd_nnet;
/* Consumes the collapsed model (nnet3/nnet-utils.cc:2006 CollapseModel output) as
 * a chain of fused layers.  Weights are copied to HBM.  frame_subsampling_factor
 * as in NnetSimpleComputationOptions (nnet-am-decodable-simple.h:62-66). The last
 * layer is the "output" node. */
kamd_nnet *kamd_nnet_create(const kamd_layer_desc *layers, int n_layers,
                            int input_dim, int frame_subsampling_factor);
void kamd_nnet_destroy(kamd_nnet *n);
int kamd_nnet_output_dim(const kamd_nnet *n);
int kamd_nnet_input_dim(const kamd_nnet *n);      /* dim of the "input" node */
int kamd_nnet_ivector_dim(const kamd_nnet *n);    /* dim of the "ivector" node, 0 without one */
int kamd_nnet_left_context(const kamd_nnet *n);   /* ComputeSimpleNnetContext */
int kamd_nnet_right_context(const kamd_nnet *n);
/* DecodableNnetSimple semantics (nnet-am-decodable-simple.cc:40-47): number of
 * output rows for T input frames = ceil(T / subsampling). */
int kamd_nnet_num_output_frames(const kamd_nnet *n, int num_input_frames);
int kamd_nnet_frame_subsampling_factor(const kamd_nnet *n);
/* Whole-batch forward, device resident.  Inputs: d_feats rows of utterance u at
 * [h_in_row_off[u], h_in_row_off[u+1]) with leading dimension ld_in; optional
 * per-utterance ivectors d_ivectors [n_utts x ivector_dim] (NULL if none).
 * Output rows of utterance u start at h_out_row_off[u] in d_out (ld_out floats).
 * Equals DecodableNnetSimple::GetOutputForFrame for every subsampled frame
 * (edge frames clamped as nnet-am-decodable-simple.cc:147-160). */
int kamd_nnet_forward_batch_device(kamd_nnet *n, const float *d_feats,
                                   const int64_t *h_in_row_off, int ld_in,
                                   const float *d_ivectors, int n_utts,
                                   float *d_out, const int64_t *h_out_row_off,
                                   int ld_out, void *stream);
/* Host convenience (one utterance): feats [T x input_dim] -> out [T' x out_dim]. */
int kamd_nnet_forward(kamd_nnet *n, const float *feats, int T,
                      const float *ivector, float *out, int out_rows_cap);
/* Streaming (DecodableAmNnetLoopedOnline, nnet3/decodable-online-looped.cc:56-240):
 * number of OUTPUT frames computable from feat_frames_ready input frames, and the rows
 * [out_first, out_first+out_count) of the log-likelihood matrix, device to device.  Values
 * equal the offline forward of the complete utterance (the right context is never guessed:
 * a frame is ready only once its context exists, or when the input is finished). */
int kamd_nnet_num_frames_ready(const kamd_nnet *n, int feat_frames_ready, int input_finished);
int kamd_nnet_forward_range(kamd_nnet *n, const float *d_feats, int ld_in, int feat_frames_ready,
                            int input_finished, int out_first, int out_count, float *d_out, int ld_out);
/* total multiply-accumulates of the last forward (for the MFMA roofline). */
double kamd_nnet_last_flops(const kamd_nnet *n);

/* ------------------------------------------------------------- component -- */
/* nnet3::Component::Propagate (nnet3/nnet-component-itf.h:130-132) for one fused layer -- the compatibility entry for
 * a host that keeps nnet3's own NnetComputer and hands single components to the device.  The shape of
 * TdnnComponent::Propagate (nnet3/nnet-tdnn-component.cc:181-212; AffineComponent :1234 is the one-offset case): `in`
 * holds in_rows consecutive time steps, out row j = sum_i W_i in[j + off_i - off_min] (+ bias, then the layer's ReLU /
 * BatchNorm / post map), in_rows - (off_max - off_min) rows.  The layer's input_layer / bypass fields are ignored (those
 * are descriptors of the graph, not of the component); ivector_dim must be 0.  Matrices are device pointers. */
typedef struct kamd_component kamd_component;
kamd_component *kamd_component_create(const kamd_layer_desc *layer);
void kamd_component_destroy(kamd_component *c);
int kamd_component_output_rows(const kamd_component *c, int in_rows);
/* returns the number of rows written (kamd_component_output_rows) or a negative status */
int kamd_component_propagate(kamd_component *c, const float *d_in, int in_rows, int ld_in, float *d_out, int ld_out,
                             void *stream);

/* ----------------------------------------------------------------- model -- */
/* final.mdl -> what the decode path consumes.  Replaces, for this path, what the binary does
 * before it decodes (nnet3bin/nnet3-latgen-faster.cc:91-104): ReadKaldiObject of the
 * TransitionModel (hmm/transition-model.cc:394-420) and the AmNnetSimple (nnet3/am-nnet-simple.cc:
 * 44-54; Nnet::Read nnet3/nnet-nnet.cc:586-628), then SetBatchnormTestMode / SetDropoutTestMode /
 * CollapseModel (nnet3/nnet-utils.cc:2006), folded into kamd_layer_desc[] together with the
 * decodable's "-log prior, * acoustic_scale" (nnet-am-decodable-simple.cc:268-271).
 * Binary models only (text ones: nnet3-am-copy --binary=true).  The graph may use TdnnComponent,
 * (NaturalGradient|Fixed)AffineComponent, LinearComponent, RectifiedLinear, BatchNorm, the
 * per-element scale / offset components, dropout / no-op, LogSoftmax, and descriptors Offset,
 * Append (slices of one producer + ReplaceIndex(ivector, t, 0)), Scale, Sum (one bypass per layer);
 * anything else is an error (NULL, kamd_last_error() names the node). */
typedef struct kamd_model kamd_model;
kamd_model *kamd_model_read(const char *path, float acoustic_scale, int frame_subsampling_factor);
void kamd_model_destroy(kamd_model *m);
/* Any out pointer may be NULL.  num_tids = TransitionModel::NumTransitionIds(). */
int kamd_model_info(const kamd_model *m, int32_t *num_layers, int32_t *input_dim, int32_t *ivector_dim,
                    int32_t *num_pdfs, int32_t *num_tids, int32_t *frame_subsampling_factor);
/* The fused layers (num_layers of them); the pointers inside stay valid until kamd_model_destroy. */
const kamd_layer_desc *kamd_model_layers(const kamd_model *m);
/* Tables indexed by transition-id, [num_tids + 1] each (index 0 unused), any may be NULL:
 * id2pdf = TransitionModel::TransitionIdToPdf (the decoder graph's ilabel -> log-likelihood
 * column), tid_phone = the phone a transition-id ENTERS (hmm-state 0, not a self-loop; 0 elsewhere:
 * what DeterminizeLatticePhonePruned inserts, lat/determinize-lattice-pruned.cc:1401-1445),
 * tid2phone = TransitionIdToPhone (endpointing, online2/online-endpoint.cc:72-93). */
int kamd_model_transition_tables(const kamd_model *m, int32_t *id2pdf, int32_t *tid_phone, int32_t *tid2phone);
/* = kamd_nnet_create(kamd_model_layers(m), ...): the weights go to HBM. */
kamd_nnet *kamd_model_create_nnet(const kamd_model *m);

/* ----------------------------------------------------------------- graph -- */
/* == fst::StdArc memory layout (OpenFst 1.6.7 ArcTpl<TropicalWeight>):
 * {int ilabel; int olabel; float weight; int nextstate}; ilabel is a 1-based
 * transition-id, 0 = epsilon (decoder/lattice-faster-decoder.cc:792,879). */
typedef struct {
  int32_t ilabel;
  int32_t olabel;
  float weight;
  int32_t nextstate;
} kamd_arc;

typedef struct kamd_graph kamd_graph;
/* HCLG as CSR in OpenFst arc order: arcs of state s are arcs[arc_off[s] ..
 * arc_off[s+1]); final_cost[s] = Final(s).Value() (+inf if not final).
 * Replaces the fst::Fst<StdArc> argument of LatticeFasterDecoderTpl's ctor
 * (decoder/lattice-faster-decoder.h:238-246) and ReadFstKaldiGeneric
 * (fstext/kaldi-fst-io.cc:44-89).  The graph is copied to HBM, split into an
 * emitting CSR and an epsilon CSR.  Epsilon cycles are rejected. */
kamd_graph *kamd_graph_create(int32_t num_states, int32_t start_state,
                              const int64_t *arc_off, const kamd_arc *arcs,
                              const float *final_cost);
/* ReadFstKaldiGeneric (fstext/kaldi-fst-io.cc:44-89): an OpenFst binary file holding a
 * "vector" or "const" FST over StdArc (HCLG.fst as the recipes write it,
 * egs/librispeech/s5/local/chain/run_tdnn_1d.sh:299-303), symbol tables skipped. */
kamd_graph *kamd_graph_read_openfst(const char *path);
/* The same file into malloc'ed host arrays (release with kamd_host_free); needs no GPU. */
int kamd_openfst_read(const char *path, int32_t *num_states, int32_t *start_state,
                      int64_t **arc_off, kamd_arc **arcs, float **final_cost);
/* fst::Fst::Write for StdArc: fst_type 0 = VectorFst, 1 = ConstFst (align != 0: the 16-byte
 * aligned variant, fst/const-fst.h). */
int kamd_openfst_write(const char *path, int fst_type, int align, int32_t num_states,
                       int32_t start_state, const int64_t *arc_off, const kamd_arc *arcs,
                       const float *final_cost);
void kamd_host_free(void *p);
void kamd_graph_destroy(kamd_graph *g);
int32_t kamd_graph_num_states(const kamd_graph *g);
int64_t kamd_graph_num_arcs(const kamd_graph *g);

/* --------------------------------------------------------------- decoder -- */
/* decoder/lattice-faster-decoder.h:38-64 LatticeFasterDecoderConfig. */
typedef struct {
  float beam;            /* 16 */
  int32_t max_active;    /* INT32_MAX */
  int32_t min_active;    /* 200 */
  float lattice_beam;    /* 10 */
  int32_t prune_interval;/* 25 */
  float beam_delta;      /* 0.5 */
  float hash_ratio;      /* 2.0 (unused on device; kept for drop-in) */
  float prune_scale;     /* 0.1 */
} kamd_decoder_config;
void kamd_decoder_config_default(kamd_decoder_config *c);

/* device sizing knobs (no reference counterpart: the reference mallocs). */
typedef struct {
  int32_t max_lanes;          /* concurrently live decoder instances */
  int32_t hash_capacity;      /* per lane, power of two, tokens of one frame */
  int64_t arena_tokens;       /* token records per lane (pool = max_lanes x this) */
  int64_t arena_links;        /* forward-link records per lane (pool likewise) */
  int32_t max_frames;         /* frames per lane */
} kamd_decoder_sizes;
void kamd_decoder_sizes_default(kamd_decoder_sizes *s);
/* Sizes for max_lanes lanes of at most max_frames decoded frames each (avg_frames on average, 0 = max_frames:
 * kamd_decoder_reserve splits the pools by utterance length), scaled down to hbm_fraction (0 = 0.5) of the
 * HBM that is free at the call.  hash_capacity / tokens_per_frame / links_per_frame: 0 = derived from
 * cfg->max_active (3 x max_active + 2000 tokens a frame, capped at 62000). */
int kamd_decoder_sizes_suggest(const kamd_decoder_config *cfg, int max_lanes, int max_frames, int avg_frames,
                               int hash_capacity, int tokens_per_frame, int links_per_frame, float hbm_fraction,
                               kamd_decoder_sizes *out);

typedef struct kamd_decoder kamd_decoder;
/* LatticeFasterDecoderTpl(const FST&, const Config&) (lattice-faster-decoder.cc:
 * 30-36).  tid2pdf[0..num_tids] is TransitionModel::id2pdf_id_
 * (hmm/transition-model.h:310,334-345; index 0 unused) so that the device reads
 * loglikes[frame][tid2pdf[ilabel]] like DecodableMatrixMapped
 * (decoder/decodable-matrix.cc:62-69).  tid2pdf == NULL means identity-1
 * (ilabel-1 indexes the loglike row directly, DecodableMatrixScaled). */
kamd_decoder *kamd_decoder_create(const kamd_graph *g,
                                  const kamd_decoder_config *cfg,
                                  const kamd_decoder_sizes *sizes,
                                  const int32_t *tid2pdf, int32_t num_tids);
void kamd_decoder_destroy(kamd_decoder *d);
int kamd_decoder_set_options(kamd_decoder *d, const kamd_decoder_config *cfg);
/* Which order-independent member of the reference's family of results the lanes compute.  The
 * reference prunes emitting arcs against a bound that tightens WHILE the tokens are visited in
 * HashList order (lattice-faster-decoder.cc:798-800), so what it keeps depends on that order.
 *   1 (default) canonical: an arc is kept iff its cost is within the bound the reference ENDS
 *     the frame with -- the tightest member; identical lattices whenever the beam alone prunes.
 *   2 canonical-loose: on the frames where max_active / min_active made the adaptive beam
 *     differ from the beam, kept iff within the bound the reference STARTS the frame with (the
 *     seed from the best token's arcs, :757-772) -- every token the reference can create in any
 *     visiting order is created; there mode 1 searches less than the reference and mode 2 at
 *     least as much.  On all other frames the tokens between the two bounds can never be
 *     expanded (next frame's cutoff = this frame's final bound) and the modes coincide. */
int kamd_decoder_set_search_mode(kamd_decoder *d, int mode);
/* Token pre-selection of the work-queue lanes (on by default; KAMD_PRESELECT=0 in the environment switches it off at
 * creation).  On a frame that records several times max_active candidate arcs (the frames behind a word boundary) the lane
 * inserts only the candidates under a bound that provably contains the next frame's max-active cutoff, plus those whose
 * target state has epsilon arcs; the tokens the other candidates would create are exactly the ones the next frame's
 * GetCutoff (lattice-faster-decoder.cc:693-702) discards unexpanded, and links from those candidates into tokens that do
 * exist are kept.  Raw lattices, best paths, per-frame cutoffs and all work counters are unchanged, EXCEPT counters[5]
 * (N_tok) and the per-frame token trace, which then count the tokens the lane inserted, not the ones the reference
 * would have created.  off = every token is created and counted, as the reference does. */
int kamd_decoder_set_token_preselection(kamd_decoder *d, int on);
/* How a lane's 160 KB of LDS were split: entries of a frame's log-likelihood row staged in LDS
 * (pdfs beyond that are read from HBM) and words of the level-1 token table (states whose
 * probe window is full spill to the level-2 table in HBM).  Diagnostic, used by the tests. */
int kamd_decoder_lds_layout(const kamd_decoder *d, int32_t *num_pdfs_lds, int32_t *table_words);
/* Words of the level-1 (LDS) token table region: a power of two up to what kamd_decoder_lds_layout reports at creation
 * (the default), or 0 for none.  A frame uses half of the region, or -- after a frame that created many tokens -- all of
 * it; states that find their probe window full go to the level-2 table in HBM.  Results do not depend on it: the tests
 * shrink the region to drive ordinary test cases through the level-2 and large-frame paths. */
int kamd_decoder_set_level1_table(kamd_decoder *d, int32_t table_words);
/* Decoder lanes that share one compute unit in this build (1: 1024-thread lanes, 2: 512-thread lanes):
 * resident lanes of a work-queue launch default to compute units x this. */
int kamd_decoder_lanes_per_cu(void);
/* Optional: split the token / link pools between lanes 0..n-1 in proportion to the
 * number of frames each will decode (utterance lengths differ 1-35 s); lanes >= n get
 * nothing.  Default is a uniform split.  Call before kamd_decoder_init. */
int kamd_decoder_reserve(kamd_decoder *d, const int32_t *lane_frames, int n);

/* One task = "advance lane L by n_frames frames of this log-likelihood matrix".
 * d_loglikes points at the row of the first frame to decode (row stride ld). */
typedef struct {
  int32_t lane;
  int32_t n_frames;
  const float *d_loglikes;
  int32_t ld;
  int32_t reserved;
} kamd_decode_task;

/* InitDecoding() (lattice-faster-decoder.cc:56-73) for the given lanes. */
int kamd_decoder_init(kamd_decoder *d, const int32_t *lanes, int n, void *stream);
/* AdvanceDecoding() (lattice-faster-decoder.cc:593-632) for a set of lanes in
 * one launch: one persistent workgroup per task.  Asynchronous. */
int kamd_decoder_advance(kamd_decoder *d, const kamd_decode_task *tasks, int n,
                         void *stream);
/* FinalizeDecoding() (lattice-faster-decoder.cc:638-653): final-cost handling and
 * the exact backward extra-cost fixpoint + pruning with lattice_beam. */
int kamd_decoder_finalize(kamd_decoder *d, const int32_t *lanes, int n,
                          void *stream);
/* Blocks until all queued work of the decoder is done; reports device-side
 * failures (arena/hash overflow) of any lane as KAMD_ERR_CAPACITY. */
int kamd_decoder_sync(kamd_decoder *d);
/* The same wait, but a lane's capacity overflow is reported per lane (lane_error[i] = flags of
 * lanes[i], 0 = fine) instead of failing the call: one stream that outgrows its arena must not
 * stop the other streams of a server.  Fails only on runtime errors. */
int kamd_decoder_sync_lanes(kamd_decoder *d, const int32_t *lanes, int n, int32_t *lane_error);

int kamd_decoder_num_frames_decoded(kamd_decoder *d, int lane);
/* FinalRelativeCost() / ReachedFinal() (lattice-faster-decoder.h:283-300). */
float kamd_decoder_final_relative_cost(kamd_decoder *d, int lane);
int kamd_decoder_reached_final(kamd_decoder *d, int lane);

/* GetRawLattice(&lat, use_final_probs=true) (lattice-faster-decoder.cc:113-196).
 * Lattice states = surviving tokens, numbered frame by frame and, inside a frame,
 * by HCLG state id (canonical; the reference numbering is pointer-order,
 * SURVEY App. D.9).  Arc weights are (graph_cost, acoustic_cost - cost_offset).
 * Two-call protocol: kamd_decoder_lattice_size, then _get with caller buffers. */
typedef struct {
  int32_t num_states;
  int32_t num_arcs;
  int32_t num_frames;
  int32_t start;        /* lattice state of the start token (frame 0) */
} kamd_lattice_size;
typedef struct {
  int32_t src, dst;      /* lattice state ids */
  int32_t ilabel, olabel;
  float graph_cost, acoustic_cost;
} kamd_lat_arc;
int kamd_decoder_lattice_size(kamd_decoder *d, int lane, kamd_lattice_size *sz);
/* state_frame[num_states], state_hclg[num_states] (HCLG state of the token),
 * state_cost[num_states] (forward tot_cost), state_final[num_states]
 * (LatticeWeight(final_cost,0) value1; +inf = not final), arcs[num_arcs] sorted
 * by (src,dst,ilabel,olabel). */
int kamd_decoder_get_raw_lattice(kamd_decoder *d, int lane, int32_t *state_frame,
                                 int32_t *state_hclg, float *state_cost,
                                 float *state_final, kamd_lat_arc *arcs);
/* GetRawLattice on a LIVE decoder (decoder/lattice-faster-decoder.cc:113-196 with !decoding_finalized_; what
 * SingleUtteranceNnet3DecoderTpl::GetLattice(end_of_utterance = false) reads between chunks, online2/online-nnet3-
 * decoding.cc:66-79): every token and forward link the lane holds, final costs computed on the spot (use_final_probs
 * = 0: every token of the last frame final with weight One).  A host-side read of the lane's arenas after
 * kamd_decoder_sync: nothing is launched, the lane decodes on.  Same canonical numbering as kamd_decoder_get_raw_lattice. */
int kamd_decoder_live_lattice_size(kamd_decoder *d, int lane, int use_final_probs, kamd_lattice_size *sz);
int kamd_decoder_get_live_raw_lattice(kamd_decoder *d, int lane, int use_final_probs, int32_t *state_frame, int32_t *state_hclg,
                                      float *state_cost, float *state_final, kamd_lat_arc *arcs);
/* PruneLattice(beam) on a raw lattice held on the host (lat/lattice-functions.cc; the exact form of what
 * LatticeFasterOnlineDecoderTpl::GetRawLatticePruned, decoder/lattice-faster-online-decoder.cc:168-265, approximates):
 * state_map[s] = number of state s in the pruned lattice or -1, arc_keep[i] = 1 for the arcs that stay. */
int kamd_lattice_prune(int32_t num_states, int32_t start, const float *state_final, const kamd_lat_arc *arcs, int32_t num_arcs,
                       float beam, int32_t *state_map, uint8_t *arc_keep, int32_t *num_states_out, int32_t *num_arcs_out);
/* GetBestPath (lattice-faster-decoder.cc:102-108) + GetLinearSymbolSequence
 * (fstext/fstext-utils-inl.h:178): ShortestPath over the raw lattice with
 * LatticeWeight ordering.  Outputs the alignment (transition-ids), words, and the
 * total (graph, acoustic) weight.  Buffers sized by *_cap; returns counts. */
int kamd_decoder_best_path(kamd_decoder *d, int lane, int32_t *alignment,
                           int ali_cap, int *ali_len, int32_t *words,
                           int words_cap, int *words_len, float *graph_cost,
                           float *acoustic_cost);
/* ------------------------------------------ work queue over resident lanes -- */
/* Test-set decoding.  NnetBatchDecoder keeps N decoder threads, each with its own
 * LatticeFasterDecoder, pulling utterances as they become ready
 * (nnet3/nnet-batch-compute.h:606-833, .cc:1156-1215 Decode()); decode.sh spreads a test
 * set over --nj jobs (steps/nnet3/decode.sh:96,123).  Here `resident_lanes` persistent
 * workgroups (one CU each) pop tasks from a device-side queue: InitDecoding,
 * AdvanceDecoding over the utterance's log-likelihood rows, FinalizeDecoding, then the
 * pruned raw lattice is copied into a device pool and a record is published in
 * host-visible memory while the kernel keeps running. */
typedef struct {
  const float *d_loglikes;   /* first row of the utterance's log-likelihoods (device) */
  int32_t ld;                /* row stride */
  int32_t n_frames;
  int32_t utt;               /* caller's utterance index: slot in the result table */
  int32_t reserved;
} kamd_queue_task;
typedef struct {
  int32_t status;            /* 0 = pending, 1 = done (written last, system-scope release) */
  int32_t error;             /* capacity flags of kamd_decoder_sync's message; 64 = lattice pool exhausted; 32 with one of
                              * 512 .. 32768 = a graph lookup met a state that is no state of HCLG and the lane stopped
                              * (which lookup: kaldi_amd/csrc/decoder.hip ERR_BAD_STATE) */
  int32_t lane, n_frames;    /* lane that decoded it; NumFramesDecoded() */
  int32_t n_tok, n_link;     /* raw lattice: states, arcs */
  int32_t n_last;            /* tokens on the last frame (their final costs travel with the lattice) */
  int32_t n_preselected;     /* frames whose inserts were pre-selected (kamd_decoder_set_token_preselection): diagnostic */
  float final_relative_cost, final_best_cost;
  int64_t blob_off, blob_bytes;   /* the lattice inside the pool */
  int64_t counters[8];       /* the work counters of kamd_decoder_get_counters */
  uint64_t phase_cycles[16];
} kamd_queue_result;
/* Lattice pool (grow-only; default 1 GiB at the first launch): page-locked host memory that the
 * queue kernel writes every finished utterance's raw lattice into, so that the host reads it
 * in place. */
int kamd_decoder_queue_configure(kamd_decoder *d, int64_t pool_bytes);
/* Asynchronous launch on `stream`: tasks are handed out in the order given (pass the
 * longest first).  resident_lanes <= max_lanes of the decoder; 0 = one per compute unit.
 * Every lane uses the decoder's per-lane arena (uniform split). */
int kamd_decoder_queue_launch(kamd_decoder *d, const kamd_queue_task *tasks, int n,
                              int resident_lanes, void *stream);
/* Non-blocking: utterance indices that finished since the last call, in completion
 * order; returns how many were written to utts[0..cap). */
int kamd_decoder_queue_poll(kamd_decoder *d, int32_t *utts, int cap);
/* The record of a finished utterance (status == 1), copied out of the shared table. */
int kamd_decoder_queue_result(kamd_decoder *d, int32_t utt, kamd_queue_result *out);
/* Raw lattice of a finished utterance: its blob (in the host-resident pool already;
 * `copy_stream` is unused and may be NULL) + canonical numbering
 * (states by (frame, HCLG state), arcs sorted), as kamd_decoder_get_raw_lattice.  All
 * output arrays are malloc'ed (kamd_host_free); thread-safe for distinct utterances. */
int kamd_decoder_queue_fetch_lattice(kamd_decoder *d, int32_t utt, void *copy_stream,
                                     int32_t *num_states, int32_t *num_arcs, int32_t *start,
                                     int32_t **state_frame, int32_t **state_hclg,
                                     float **state_cost, float **state_final, kamd_lat_arc **arcs);
/* Blocks until the queue kernel has ended; *ms = its duration (HIP events on the launch
 * stream), *lanes = resident lanes used. */
/* The same for FEW utterances that need MUCH room: n <= max_lanes tasks on n lanes, the token and link pools split between
 * just these n (each lane's arenas are max_lanes / n times the usual ones) -- the second chance NnetBatchDecoder gives an
 * utterance whose lane ran out of arena space.  Nothing of this decoder may be in flight; the next ordinary launch
 * restores the uniform split.  Poll / result / fetch / wait as above. */
int kamd_decoder_queue_launch_wide(kamd_decoder *d, const kamd_queue_task *tasks, int n, void *stream);
/* kamd_decoder_sizes.max_lanes of the object. */
int kamd_decoder_max_lanes(const kamd_decoder *d);
/* ... and the frames per utterance its per-lane arrays were sized for (kamd_decoder_sizes.max_frames). */
int kamd_decoder_max_frames(const kamd_decoder *d);
int kamd_decoder_queue_wait(kamd_decoder *d, float *ms, int32_t *lanes);
/* GetBestPath over a raw lattice given as arrays (the same ShortestPath as
 * kamd_decoder_best_path; no decoder state, thread-safe). */
int kamd_lattice_best_path(int32_t num_states, int32_t start, const float *state_final,
                           const kamd_lat_arc *arcs, int32_t num_arcs, int32_t *alignment,
                           int ali_cap, int *ali_len, int32_t *words, int words_cap,
                           int *words_len, float *graph_cost, float *acoustic_cost);

/* LatticeWriter entry (util/kaldi-table TableWriter + LatticeHolder::Write,
 * lat/kaldi-lattice.h:75-118; WriteLattice lat/kaldi-lattice.cc:96-130): appends
 * "key " + the lattice, binary (OpenFst VectorFst of "lattice4" arcs, no Kaldi binary
 * marker) or text (FstPrinter lines between two newlines).  state_final[2s], [2s+1] =
 * final LatticeWeight (graph, acoustic); graph = +inf means not final.  arcs sorted by
 * src (as kamd_decoder_get_raw_lattice returns them).  What nnet3-latgen-faster writes
 * with --determinize-lattice=false (decoder/decoder-wrappers.cc:251-262). */
int kamd_lattice_write(const char *path, int append, const char *key, int binary,
                       int32_t num_states, int32_t start, const float *state_final,
                       const kamd_lat_arc *arcs, int32_t num_arcs);
/* Next entry of a lattice archive at byte *offset (advanced past it); either form is
 * accepted (first byte 214 = binary, lat/kaldi-lattice.cc:366-386).  Returns 1 at end of
 * file.  state_final / arcs are malloc'ed: kamd_host_free. */
int kamd_lattice_read(const char *path, int64_t *offset, char *key, int key_cap,
                      int32_t *num_states, int32_t *start, float **state_final,
                      kamd_lat_arc **arcs, int32_t *num_arcs);
/* ----------------------------------------------------- input-side formats -- */
/* WaveData::Read (feat/wave-reader.cc:113-318): RIFF / RIFX 16-bit PCM, filler chunks,
 * WAVE_FORMAT_EXTENSIBLE, streamed sizes.  *data: num_channels rows of num_samples floats in
 * int16 range (malloc'ed: kamd_host_free). */
int kamd_wave_read(const char *path, float *samp_freq, int32_t *num_channels, int64_t *num_samples,
                   float **data);
/* Next entry of a Kaldi float-matrix archive ("ark": key, space, object) at byte *offset
 * (advanced past it; key == NULL: the object itself starts at *offset, as an scp line's
 * "file:offset" points): binary FM / DM, compressed CM / CM2 / CM3 or text
 * (matrix/kaldi-matrix.cc:1378-1512, matrix/compressed-matrix.cc:566-650).  Returns 1 at end of
 * file.  *data row-major [rows x cols], malloc'ed.  A float VECTOR entry (binary FV / DV, text
 * " [ 1 2 3 ]": BaseFloatVectorHolder, e.g. an --ivectors archive) comes back as one row. */
int kamd_ark_read_matrix(const char *path, int64_t *offset, char *key, int key_cap, int32_t *rows,
                         int32_t *cols, float **data);
int kamd_ark_write_matrix(const char *path, int append, const char *key, int binary, int32_t rows,
                          int32_t cols, const float *data);
/* Next Int32VectorHolder entry (alignments, word sequences, a dumped transition-id -> pdf table) */
int kamd_ark_read_int32_vector(const char *path, int64_t *offset, char *key, int key_cap, int32_t *n,
                               int32_t **data);

/* ------------------------------------------------ GMM acoustic model (configs[0]) -- */
/* AmDiagGmm (gmm/am-diag-gmm.h:40-120) behind DecodableAmDiagGmmScaled (gmm/decodable-am-diag-gmm.h:50-120, .cc:27-70):
 * pdf p owns Gaussians [mix_off[p], mix_off[p+1]) of gconsts [G], means_invvars / inv_vars [G x dim] (DiagGmm's members).
 * loglikes: out[t][p] = scale * LogSumExp over the pdf's Gaussians, a [rows x num_pdfs] matrix the decoder takes as a
 * DecodableMatrixMapped with the transition model's id2pdf. */
typedef struct kamd_am_gmm kamd_am_gmm;
kamd_am_gmm *kamd_am_gmm_create(int32_t num_pdfs, int32_t dim, const int32_t *mix_off, const float *gconsts,
                                const float *means_invvars, const float *inv_vars);
void kamd_am_gmm_destroy(kamd_am_gmm *g);
int kamd_am_gmm_num_pdfs(const kamd_am_gmm *g);
int kamd_am_gmm_dim(const kamd_am_gmm *g);
int kamd_am_gmm_loglikes_device(kamd_am_gmm *g, const float *d_feats, int ld, int64_t rows, float scale, float *d_out, void *stream);
int kamd_am_gmm_loglikes(kamd_am_gmm *g, const float *feats, int num_frames, int feat_dim, float scale, float *out);

/* -------- feature post-processing: CMVN, add-deltas, splice / transform -- */
/* splice-feats | transform-feats (featbin/splice-feats.cc, transform-feats.cc:100-160): SpliceFrames with left / right
 * context clamped at each utterance's ends, then y = M x, or M [x; 1] when M has one more column (LDA+MLLT final.mat,
 * per-speaker fMLLR).  h_transforms: n_transforms matrices [xf_rows x xf_cols] back to back, utterance u uses
 * h_utt_transform[u]; h_transforms == NULL: the spliced vectors are the output (left = right = 0 with a transform:
 * transform-feats alone). */
int kamd_feat_splice_transform_device(const float *d_in, int ld_in, float *d_out, int ld_out, const int64_t *h_row_off,
                                      int n_utts, int dim, int left, int right, const float *h_transforms,
                                      int n_transforms, const int32_t *h_utt_transform, int xf_rows, int xf_cols,
                                      void *stream);
/* add-deltas: ComputeDeltas / DeltaFeatures::Process (feat/feature-functions.cc:118-165, featbin/add-deltas.cc): every
 * utterance's [T x dim] features -> [T x (order+1)*dim] (orders 0..order, window frames each side, clamped at the
 * utterance's ends). */
int kamd_feat_add_deltas_device(const float *d_in, int ld_in, float *d_out, int ld_out, const int64_t *h_row_off,
                                int n_utts, int dim, int order, int window, void *stream);
/* compute-cmvn-stats: AccCmvnStats (transform/cmvn.cc:30-62) over every utterance of a batch of device
 * features; utterance u owns rows [row_off[u], row_off[u+1]).  h_stats: n_utts x [2 x (dim+1)] doubles
 * (sums | count, sums of squares | unused), ACCUMULATED into (zero it for fresh statistics; pass a
 * speaker's running statistics to add an utterance to them). */
int kamd_cmvn_acc_stats_device(const float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts,
                               double *h_stats, void *stream);
/* compute-cmvn-stats --weights: AccCmvnStats(feats, &weights, stats) (transform/cmvn.cc:49-62); d_weights holds one
 * float per row of the batch (device), frames of weight 0 are skipped, the count is the sum of the weights. */
int kamd_cmvn_acc_stats_weighted_device(const float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts,
                                        const float *d_weights, double *h_stats, void *stream);
/* apply-cmvn: ApplyCmvn (transform/cmvn.cc:64-118) in place, utterance u with statistics h_stats[u]
 * (its own, its speaker's, or global ones).  norm_means = 0 leaves the features unchanged, as the
 * binary does; norm_vars without norm_means is an error (featbin/apply-cmvn.cc:63-64). */
int kamd_cmvn_apply_device(float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts,
                           const double *h_stats, int norm_means, int norm_vars, void *stream);
/* apply-cmvn --reverse: ApplyCmvnReverse (transform/cmvn.cc:120-168), same arguments: zero-mean (unit-variance)
 * features get the statistics' mean (and variance) back. */
int kamd_cmvn_apply_reverse_device(float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts,
                           const double *h_stats, int norm_means, int norm_vars, void *stream);

/* --------------------------------------------------- online i-vector extraction -- */
/* OnlineIvectorFeature as ivector-extract-online2 drives it (online2/online-ivector-feature.cc:
 * 150-420, online2bin/ivector-extract-online2.cc:95-175; use_most_recent_ivector = false, no frame
 * weights, a fresh adaptation state per utterance): base features -> OnlineCmvn (sliding window
 * smoothed with the global stats) -> splice -> LDA for the UBM posteriors; splice -> LDA of the
 * raw features for the statistics; diagonal-UBM log-likelihoods -> VectorToPosteriorEntry
 * (hmm/posterior.cc:440-508) scaled by posterior_scale; OnlineIvectorEstimationStats::AccStats
 * (ivector/ivector-extractor.cc:611-668); every ivector_period frames num_cg_iters steps of
 * LinearCgd (matrix/optimization.cc:453-557) from the previous estimate.  Row i of the result is
 * the estimate after frames 0 .. i * ivector_period, minus the prior offset in dimension 0: what
 * nnet3-latgen-faster takes as --online-ivectors. */
typedef struct kamd_ivector_extractor kamd_ivector_extractor;
typedef struct kamd_ivector_desc {
  int32_t feat_dim;                 /* base features (e.g. 40 hires MFCCs) */
  int32_t splice_left, splice_right;/* OnlineSpliceOptions (feat/online-feature.h:446-456) */
  int32_t lda_rows, lda_cols;       /* final.mat: cols = feat_dim*(left+1+right) [+1: offset column] */
  const float *lda;                 /* [lda_rows x lda_cols] row-major */
  const double *global_cmvn_stats;  /* [2 x (feat_dim+1)]: sums, counts; row 1 = sums of squares */
  int32_t cmn_window, speaker_frames, global_frames;   /* OnlineCmvnOptions (feat/online-feature.h:200-230) */
  int32_t normalize_mean, normalize_variance;          /* OnlineCmvnOptions (feat/online-feature.h:170-215); variance needs the mean */
  int32_t num_gauss;                /* diagonal UBM over the lda_rows-dimensional features (final.dubm) */
  const float *ubm_gconsts;         /* [num_gauss] */
  const float *ubm_means_invvars;   /* [num_gauss x lda_rows] */
  const float *ubm_inv_vars;        /* [num_gauss x lda_rows] */
  int32_t ivector_dim;              /* <= 128 */
  const double *M;                  /* IvectorExtractor::M_: [num_gauss][lda_rows][ivector_dim] */
  const double *sigma_inv;          /* Sigma_inv_: [num_gauss][lda_rows*(lda_rows+1)/2] packed lower triangle */
  double prior_offset;
  int32_t ivector_period, num_gselect, num_cg_iters;   /* 10, 5, 15 */
  float min_post, posterior_scale, max_count;          /* 0.025, 0.1, 0 */
} kamd_ivector_desc;
kamd_ivector_extractor *kamd_ivector_extractor_create(const kamd_ivector_desc *desc);
/* The files of an i-vector extraction config -- what OnlineIvectorExtractionInfo::Init reads (online2/online-ivector-
 * feature.cc:30-74) for --ivector-extraction-config of the online2 binaries / --config of ivector-extract-online2:
 * final.mat, global_cmvn.stats, the OnlineCmvnOptions and splice config files, final.dubm (gconsts recomputed as
 * DiagGmm::Read does), final.ie.  Binary objects; rxfilenames may be pipes or file:offset.  The descriptor's pointers
 * stay valid until kamd_ivector_info_destroy.  NULL + kamd_last_error() on a missing option / file / mismatch. */
typedef struct kamd_ivector_info kamd_ivector_info;
kamd_ivector_info *kamd_ivector_info_read(const char *config_rxfilename);
void kamd_ivector_info_destroy(kamd_ivector_info *info);
const kamd_ivector_desc *kamd_ivector_info_desc(const kamd_ivector_info *info);
kamd_ivector_extractor *kamd_ivector_info_create_extractor(const kamd_ivector_info *info);
void kamd_ivector_extractor_destroy(kamd_ivector_extractor *e);
int kamd_ivector_dim(const kamd_ivector_extractor *e);
int kamd_ivector_period(const kamd_ivector_extractor *e);
/* rows of the result for an utterance of num_frames frames: ceil(num_frames / period) */
int kamd_ivector_num_ivectors(const kamd_ivector_extractor *e, int num_frames);
/* Batch form: utterance u owns rows [h_row_off[u], h_row_off[u+1]) of the device feature matrix
 * (leading dimension ld_feat >= feat_dim) and rows [h_out_row_off[u], ...) of d_out
 * [rows x ivector_dim]. */
int kamd_ivector_extract_online_device(kamd_ivector_extractor *e, const float *d_feats, const int64_t *h_row_off,
                                       int ld_feat, int n_utts, float *d_out, const int64_t *h_out_row_off,
                                       void *stream);
/* The same in two halves, for a driver that walks a test set pass by pass (kamd_batch_decoder does): OnlineIvectorFeature's
 * UpdateStatsUntilFrame (online2/online-ivector-feature.cc:322-364: AccStats of every frame, what does not depend on the
 * order of the steps) for the utterances of one pass, and -- once, over ALL passes' utterances -- the per-step
 * GetIvector chain (ivector-extractor.cc:732-756, 15 LinearCgd iterations per step, one sequential chain per utterance:
 * a launch costs its longest utterance whatever else there is, so one launch instead of one per pass).  h_out_row_off numbers
 * the utterances' rows in the whole set's i-vector matrix, [0, total_iv_rows); the step statistics (5151 doubles per row)
 * stay on the device between the calls.  reserve -> stats (per pass) -> solve; the results equal the one-call form bit for bit. */
int kamd_ivector_online_reserve_steps(kamd_ivector_extractor *e, int64_t total_iv_rows);
int kamd_ivector_online_stats_device(kamd_ivector_extractor *e, const float *d_feats, const int64_t *h_row_off, int ld_feat,
                                     int n_utts, const int64_t *h_out_row_off, void *stream);
int kamd_ivector_online_solve_device(kamd_ivector_extractor *e, const int64_t *h_row_off, int n_utts, float *d_out,
                                     const int64_t *h_out_row_off, void *stream);
/* The same with the speaker's adaptation state (OnlineIvectorExtractorAdaptationState, SetAdaptationState /
 * GetAdaptationState, online2/online-ivector-feature.cc:400-435), as ivector-extract-online2 carries it
 * from one utterance of a speaker to the next (online2bin/ivector-extract-online2.cc:95-170).  A state is
 * kamd_ivector_state_size() doubles: [2 x (feat_dim+1)] speaker CMVN stats, the packed quadratic term,
 * the linear term, the frame count.  h_state_in / h_state_out: n_utts states on the host, either may
 * be NULL (fresh state / not wanted); the utterances of one call must belong to different speakers.
 * The state returned is the one BEFORE LimitFrames: apply kamd_ivector_state_limit_frames (host
 * arithmetic) with --max-remembered-frames before handing it to the speaker's next utterance. */
int kamd_ivector_state_size(const kamd_ivector_extractor *e);
int kamd_ivector_extract_online_adapt_device(kamd_ivector_extractor *e, const float *d_feats, const int64_t *h_row_off,
                                             int ld_feat, int n_utts, float *d_out, const int64_t *h_out_row_off,
                                             const double *h_state_in, double *h_state_out, void *stream);
int kamd_ivector_state_limit_frames(const kamd_ivector_extractor *e, double *state, float max_remembered_frames);
/* The extractor's row-indexed workspaces (running sums, LDA outputs, posteriors of every feature row) as an object of
 * their own.  The silence-weighted update re-reads the LDA rows of frames it processed on EARLIER calls, so a driver of
 * such a sequence (kamd_stream_batch_* does this) owns one and binds it around its calls; with none bound (NULL) the
 * extractor's own buffers serve, which is all the stateless entry points need.  Binding is not thread-safe: like every
 * call on an extractor it belongs to the thread that drives it. */
typedef struct kamd_ivector_workspace kamd_ivector_workspace;
kamd_ivector_workspace *kamd_ivector_workspace_create(void);
void kamd_ivector_workspace_destroy(kamd_ivector_workspace *w);
int kamd_ivector_extractor_bind_workspace(kamd_ivector_extractor *e, kamd_ivector_workspace *w);
/* Streaming form: OnlineIvectorFeature::GetFrame with use_most_recent_ivector = true (online2/online-ivector-
 * feature.cc:206-320), for n streams at once.  Stream item i has h_n_base[i] base-feature frames so far at rows
 * h_feat_row[i].. of d_feats (the extractor's workspaces mirror that row space: ws_rows_total rows); frames
 * [h_n_done[i], h_n_upto[i]) are new (h_n_upto <= h_n_base - splice_right unless the input is finished): they enter
 * the statistics as one batch, then num_cg_iters CG steps run from the stream's current estimate.  A stream's
 * record = kamd_ivector_stream_record_size() doubles on the device (adaptation state | current estimate),
 * started from kamd_ivector_stream_record_init and updated in place; h_record[i] = index of item i's record in
 * d_records.  d_out row i = the estimate, prior offset removed from dimension 0.  The speaker CMVN part of the
 * record is NOT advanced (add the utterance with kamd_cmvn_acc_stats_device when it ends). */
int kamd_ivector_stream_record_size(const kamd_ivector_extractor *e);
int kamd_ivector_stream_record_init(const kamd_ivector_extractor *e, const double *state, double *record);
int kamd_ivector_stream_update_device(kamd_ivector_extractor *e, const float *d_feats, int ld_feat, int64_t ws_rows_total,
                                      const int64_t *h_feat_row, const int32_t *h_n_base, const int32_t *h_n_done,
                                      const int32_t *h_n_upto, const int32_t *h_record, int n, double *d_records,
                                      float *d_out, void *stream);
/* The same tick with silence weighting (UpdateStatsUntilFrameWeighted + UpdateStatsForFrames, online2/online-ivector-
 * feature.cc:191-227, 263-306): frames [h_n_done[i], h_n_upto[i]) are new (their LDA features are computed and stay in
 * the workspace for later re-weighting); the statistics take item i's entries [h_wl_off[i], h_wl_off[i+1]) of
 * (h_wl_frame, h_wl_weight) -- kamd_silence_weighting_pop_until's output: increasing frames < h_n_upto[i], non-zero
 * weights -- each with min_post = GetMinPost(weight) and posteriors scaled by posterior_scale * weight.  An empty
 * list still runs GetIvector. */
int kamd_ivector_stream_update_weighted_device(kamd_ivector_extractor *e, const float *d_feats, int ld_feat,
                                               int64_t ws_rows_total, const int64_t *h_feat_row, const int32_t *h_n_base,
                                               const int32_t *h_n_done, const int32_t *h_n_upto, const int32_t *h_record,
                                               const int32_t *h_wl_off, const int32_t *h_wl_frame, const float *h_wl_weight,
                                               int n, double *d_records, float *d_out, void *stream);
/* one utterance, host in / host out; returns the number of rows written or < 0 */
int kamd_ivector_extract_online(kamd_ivector_extractor *e, const float *feats, int num_frames, float *out,
                                int out_rows_cap);
int kamd_ivector_extract_online_adapt(kamd_ivector_extractor *e, const float *feats, int num_frames, float *out,
                                      int out_rows_cap, const double *state_in, double *state_out);
/* diagnostic: the per-frame posteriors of the last batch (frame-major, num_gselect slots per frame,
 * gaussian -1 = empty slot) */
int kamd_ivector_last_posteriors(kamd_ivector_extractor *e, int32_t *gauss, float *weight, int64_t frames_cap);

/* ------------------------------------- extended filenames and table specifiers -- */
/* ClassifyRxfilename / ClassifyWxfilename (util/kaldi-io.cc:85-186); values follow the
 * reference's InputType / OutputType enums (util/kaldi-io.h:89-111). */
enum { KAMD_RX_NONE = 0, KAMD_RX_FILE = 1, KAMD_RX_STDIN = 2, KAMD_RX_OFFSET_FILE = 3, KAMD_RX_PIPE = 4 };
enum { KAMD_WX_NONE = 0, KAMD_WX_FILE = 1, KAMD_WX_STDOUT = 2, KAMD_WX_PIPE = 3 };
int kamd_classify_rxfilename(const char *filename);
int kamd_classify_wxfilename(const char *filename);
/* ClassifyRspecifier (util/kaldi-table.cc:225-310): 0 = not an rspecifier, 1 = "ark:", 2 = "scp:";
 * option letters o / s / cs / p / bg as bits.  -1 if the buffer is too small. */
enum { KAMD_RSPEC_ONCE = 1, KAMD_RSPEC_SORTED = 2, KAMD_RSPEC_CALLED_SORTED = 4, KAMD_RSPEC_PERMISSIVE = 8,
       KAMD_RSPEC_BACKGROUND = 16 };
int kamd_classify_rspecifier(const char *rspecifier, char *rxfilename, int cap, int *opts);
/* ClassifyWspecifier (util/kaldi-table.cc:115-222): 0 none, 1 "ark:", 2 "scp:", 3 "ark,scp:a,b";
 * binary defaults to set ("t" clears it). */
enum { KAMD_WSPEC_BINARY = 1, KAMD_WSPEC_FLUSH = 2, KAMD_WSPEC_PERMISSIVE = 4 };
int kamd_classify_wspecifier(const char *wspecifier, char *archive_wxfilename, int ark_cap,
                             char *script_wxfilename, int scp_cap, int *opts);
/* Any rxfilename -> a seekable (path, offset) the readers above take: "file", "file:offset",
 * "command |" and "-" (the last two spooled to a temporary file: *is_temp = 1, the caller
 * unlinks it).  Input::Open's dispatch (util/kaldi-io.cc:760-800). */
int kamd_rx_materialize(const char *rxfilename, char *path, int cap, int64_t *offset, int *is_temp);


/* ------------------------------------------------------- batched streaming -- */
/* N concurrent SingleUtteranceNnet3Decoder streams (online2/online-nnet3-decoding.{h,cc})
 * driven together: stream s is decoder lane s of `dec` (created with max_lanes >= max_streams).
 * Waveforms and features of all streams live in two pooled HBM buffers; one tick costs one
 * feature launch, one batched nnet forward and one AdvanceKernel launch, whatever the number of
 * streams.  Results per stream through the decoder entry points with lane = stream
 * (kamd_decoder_partial_best_path while streaming; kamd_decoder_finalize + lattice at the end). */
typedef struct kamd_stream_batch kamd_stream_batch;
kamd_stream_batch *kamd_stream_batch_create(kamd_feat *feat, kamd_nnet *nnet, kamd_decoder *dec,
                                            int max_streams, float max_seconds, float samp_freq);
void kamd_stream_batch_destroy(kamd_stream_batch *b);
/* new utterance on these streams (InitDecoding, online-nnet3-decoding.cc:40) */
int kamd_stream_batch_start(kamd_stream_batch *b, const int32_t *streams, int n);
/* AcceptWaveform (+ InputFinished when input_finished != 0); host samples, int16 range */
int kamd_stream_batch_accept(kamd_stream_batch *b, int stream, const float *wave, int64_t n,
                             int input_finished);
/* AcceptWaveform for many streams at once: waves[offsets[i] .. offsets[i+1]) is appended to streams[i] (each stream at most
 * once); one host-to-device copy and one scatter launch for the whole tick.  input_finished: [n] flags or NULL. */
int kamd_stream_batch_accept_many(kamd_stream_batch *b, const int32_t *streams, int n, const float *waves,
                                  const int64_t *offsets, const int32_t *input_finished);
/* AdvanceDecoding for all listed streams; frames_decoded[n] may be NULL */
int kamd_stream_batch_advance(kamd_stream_batch *b, const int32_t *streams, int n, int32_t *frames_decoded);
int kamd_stream_batch_num_frames_ready(const kamd_stream_batch *b, int stream);
/* (status bit) the silence-weighted i-vector statistics of the stream lost a batch of delta weights when a tick failed
 * between the traceback and the statistics update: the stream must be restarted. */
#define KAMD_STREAM_WEIGHTING_LOST (1 << 30)
/* Per-stream health: 0 = fine, otherwise the decoder capacity flags that took this stream out
 * (its lane's arena / table overflowed).  kamd_stream_batch_advance reports such a failure with
 * KAMD_ERR_CAPACITY but leaves every OTHER stream of the tick advanced and usable; the failed
 * stream refuses further ticks until kamd_stream_batch_start restarts it. */
int kamd_stream_batch_get_status(const kamd_stream_batch *b, const int32_t *streams, int n, int32_t *status);
/* Online i-vectors in the streaming path (online2-wav-nnet3-latgen-faster with an ivector-extraction config):
 * OnlineIvectorFeature with use_most_recent_ivector feeding DecodableNnetLoopedOnline.  Call before any stream is
 * started.  frames_per_chunk = the looped decodable's --frames-per-chunk (20 in the online recipes; its i-vector
 * period), splice_right = the extractor's splicing right context.  With it the network is served chunk by chunk
 * like the reference (chunk k once (k+1)*frames_per_chunk + right-context frames exist, or at the end); a tick
 * that makes chunks computable advances the stream's estimate once, to the most recent frame
 * (decodable-online-looped.cc:160-190), and the i-vector slots floor(t / frames_per_chunk) those chunks bring in
 * (nnet3/nnet-compile-looped.cc:186-207) get that estimate. */
int kamd_stream_batch_set_ivector_extractor(kamd_stream_batch *b, kamd_ivector_extractor *e, int frames_per_chunk,
                                            int splice_right);
/* --ivector-silence-weighting.* (online2-wav-nnet3-latgen-faster.cc:214-216, 258-266; OnlineSilenceWeighting):
 * before every tick's AdvanceDecoding the streams' tracebacks re-decide which frames are silence and the i-vector
 * statistics are corrected by the difference.  tid_is_silence[tid] = TransitionIdToPhone(tid) is a --silence-phones
 * phone; silence_weight == 1 = off.  After kamd_stream_batch_set_ivector_extractor, before any stream starts. */
int kamd_stream_batch_set_silence_weighting(kamd_stream_batch *b, const uint8_t *tid_is_silence, int n_tids,
                                            float silence_weight, float max_state_duration);
/* Arena compaction (kamd_decoder_compact = PruneActiveTokens) for streams: a stream whose token or link arena is fuller
 * than `fraction` at the start of a tick is compacted before it advances; default 0.5, 0 = never.  Bounds a long
 * utterance's memory by its pruned lattice; results do not change. */
int kamd_stream_batch_set_compaction(kamd_stream_batch *b, float fraction);
/* LatticeFasterDecoderConfig::prune_interval for streams (decoder/lattice-faster-decoder.cc:617-619: PruneActiveTokens every
 * prune_interval frames while decoding): a stream that decoded `frames` frames since its last compaction is compacted at the
 * start of its next tick, so that FinalizeDecoding at the end of the utterance finds all but the last frames pruned already
 * -- the end-of-utterance latency of a streaming host (online2bin/online2-wav-nnet3-latgen-faster.cc:264-285).  0 (default):
 * never; the final lattice and 1-best do not depend on it, a LIVE raw lattice read between ticks then holds the pruned
 * tokens of the compacted frames (as the reference's does after its own PruneActiveTokens). */
int kamd_stream_batch_set_prune_interval(kamd_stream_batch *b, int frames);
int64_t kamd_stream_batch_num_compactions(const kamd_stream_batch *b);
/* start with the speakers' adaptation states (n x kamd_ivector_state_size() doubles; NULL = fresh) and read a
 * stream's state back after its utterance (before LimitFrames) */
int kamd_stream_batch_start_adapted(kamd_stream_batch *b, const int32_t *streams, int n, const double *states);
int kamd_stream_batch_get_adaptation_state(kamd_stream_batch *b, int stream, double *state);
int kamd_stream_batch_get_ivector_slots(kamd_stream_batch *b, int stream, float *out, int rows_cap, int *first_slot,
                                        int *count);
/* building blocks of the above: frames [first, first+count) of n waveforms that are not
 * adjacent in memory, and an nnet forward over n non-adjacent feature slices */
int kamd_feat_compute_ranges_device(kamd_feat *f, const float *d_waves, const int64_t *h_wave_start,
                                    const int64_t *h_wave_len, const int32_t *h_first_frame,
                                    const int32_t *h_num_frames, int n, float *d_out,
                                    const int64_t *h_row_off, int ld_out, void *stream);
int kamd_nnet_forward_slices_device(kamd_nnet *n, const float *d_feats, const int64_t *h_in_start,
                                    const int32_t *h_in_len, int ld_in, const float *d_ivectors, int n_items,
                                    float *d_out, const int64_t *h_out_row_off, int ld_out, void *stream);
/* The same with the looped decodable's i-vectors (DecodableNnetLoopedOnline: ModifyNnetIvectorPeriod makes
 * the network read the ivector input at Round(t, period), nnet3/nnet-compile-looped.cc:164-207): the
 * first layer's row at absolute time t of item i uses row slot_base[i] + clamp(floor(t / period) -
 * slot_first[i], 0, slot_count[i] - 1) of d_ivector_table [table_rows x ivector_dim]; abs_t0[i] = absolute
 * time of the item's first input frame. */
int kamd_nnet_forward_slices_slots_device(kamd_nnet *n, const float *d_feats, const int64_t *h_in_start,
                                          const int32_t *h_in_len, int ld_in, const float *d_ivector_table,
                                          int table_rows, int period, const int32_t *h_slot_base,
                                          const int32_t *h_slot_first, const int32_t *h_slot_count,
                                          const int32_t *h_abs_t0, int n_items, float *d_out,
                                          const int64_t *h_out_row_off, int ld_out, void *stream);
/* DecodableNnetSimple with ONLINE ivectors (--online-ivectors / --online-ivector-period of
 * nnet3-latgen-faster, nnet3/nnet-am-decodable-simple.cc:93-214) for a batch of utterances:
 * the output is computed chunk by chunk (frames_per_chunk input frames, rounded up to a
 * multiple of the subsampling factor as CheckAndFixConfigs does, :278-310), every chunk with
 * the ivector row GetCurrentIvector picks for its middle (:181-211) and with its own context,
 * clamped at the utterance edges only.  d_online_ivectors: [rows x iv_dim] on the device,
 * utterance u owns rows [h_iv_row_off[u], h_iv_row_off[u+1]) (one row per ivector_period
 * frames).  All chunks of all utterances are items of one batched forward. */
int kamd_nnet_forward_chunked_device(kamd_nnet *n, const float *d_feats, const int64_t *h_in_row_off, int ld_in,
                                     const float *d_online_ivectors, const int64_t *h_iv_row_off, int iv_dim,
                                     int ivector_period, int frames_per_chunk, int n_utts, float *d_out,
                                     const int64_t *h_out_row_off, int ld_out, void *stream);
/* One minibatch of NnetInferenceTasks (nnet3/nnet-batch-compute.h:42-110) -- what NnetBatchComputer::Compute evaluates
 * (nnet-batch-compute.cc:398-470).  For a host that keeps the reference's scheduler (AcceptTask, priorities, full / partial
 * minibatches, the semaphores) and hands the device the tasks it picked: task i = output frames [first_output_t,
 * + num_output_frames) at the subsampled rate of the utterance whose features are rows [in_row, + in_len) of d_feats (the
 * context beyond the utterance's ends is its first / last frame repeated, as SplitInputToTasks pads it), evaluated with row
 * iv_row of d_ivectors (-1: the model has no i-vector input).  The tasks' outputs lie back to back in d_out.  The tasks of
 * one call need not share a shape (the device compiles no computation per shape). */
typedef struct kamd_inference_task {
  int64_t in_row;
  int32_t in_len, first_output_t, num_output_frames, iv_row;
} kamd_inference_task;
int kamd_nnet_forward_inference_tasks_device(kamd_nnet *n, const float *d_feats, int ld_in, const float *d_ivectors, int iv_dim,
                                             const kamd_inference_task *tasks, int n_tasks, float *d_out, int ld_out,
                                             void *stream);
/* The same batch evaluated the way NnetBatchComputer does (nnet3/nnet-batch-compute.h:207; SplitUtteranceIntoTasks
 * nnet-batch-compute.cc:774-829 with GetOutputFrameInfoForTasks :586-668, AddOnlineIvectorsToTasks :670-703,
 * SplitInputToTasks :705-770, then Compute and MergeTaskOutput :832-870 -- i.e. what nnet3-latgen-faster-batch, the binary
 * behind `steps/nnet3/decode.sh --use-gpu true`, feeds its decoders): tasks of frames_per_chunk / subsampling output
 * frames (integer division: 16 for 50 / 3, where DecodableNnetSimple takes 17), the last task ending on the utterance's
 * last frame and overlapping the one before it, an utterance shorter than a task being one task; every task with the
 * i-vector row of ITS middle, the last row when that is at most 20 input frames beyond the table.  The default options
 * (--extra-left-context 0, --ensure-exact-final-context false).  Every task is an item of one batched forward that
 * computes the rows MergeTaskOutput keeps; the reference's minibatch scheduling (priorities, partial minibatches) has no
 * counterpart: all tasks of all utterances run in one pass. */
int kamd_nnet_forward_tasks_device(kamd_nnet *n, const float *d_feats, const int64_t *h_in_row_off, int ld_in,
                                   const float *d_online_ivectors, const int64_t *h_iv_row_off, int iv_dim,
                                   int ivector_period, int frames_per_chunk, int n_utts, float *d_out,
                                   const int64_t *h_out_row_off, int ld_out, void *stream);

/* ------------------------------------------------- lattice determinization -- */
/* DeterminizeLatticePhonePrunedOptions + DeterminizeLatticePrunedOptions
 * (lat/determinize-lattice-pruned.h:126-141, 214-245), same defaults.  minimize is not
 * offered (the recipes leave it false). */
typedef struct {
  float delta;               /* 2^-10: weight tolerance when two subsets are compared */
  int32_t max_mem;           /* 50000000 bytes (approximate accounting); <= 0: unlimited */
  int32_t phone_determinize; /* 1: first pass on phone + word labels */
  int32_t word_determinize;  /* 1: second pass on word labels */
  int32_t max_loop;          /* 0 = no limit on epsilon-closure iterations */
  float retry_cutoff;        /* 0.5 */
} kamd_determinize_opts;
void kamd_determinize_opts_default(kamd_determinize_opts *o);
/* One arc of a CompactLattice (acceptor: label = word): weight (graph, acoustic) plus the
 * transition-id string strings[str_begin .. str_begin + str_len). */
typedef struct {
  int32_t src, dst, label;
  float graph_cost, acoustic_cost;
  int32_t str_begin, str_len;
} kamd_clat_arc;
typedef struct kamd_compact_lattice kamd_compact_lattice;
/* DeterminizeLatticePhonePrunedWrapper (lat/determinize-lattice-pruned.cc:1484-1509) on a
 * raw lattice as kamd_decoder_get_raw_lattice / kamd_lattice_read return it (ilabel =
 * transition-id, olabel = word; state_final[2s], [2s+1]; arcs sorted by src): Invert, TopSort,
 * ArcSort, phone pass, word pass, Connect.  tid_phone[tid] (tid in 1..num_tids) is the phone
 * of a transition-id that ENTERS a phone (TransitionIdToHmmState == 0 and not a self-loop,
 * :1316-1318) and 0 for every other transition-id; may be NULL when phone_determinize = 0.
 * beam = config.lattice_beam (decoder/decoder-wrappers.cc:272-277).  Host code, no GPU. */
kamd_compact_lattice *kamd_lattice_determinize_phone_pruned(
    int32_t num_states, int32_t start, const float *state_final, const kamd_lat_arc *arcs,
    int32_t num_arcs, const int32_t *tid_phone, int32_t num_tids, double beam,
    const kamd_determinize_opts *opts);
void kamd_compact_lattice_destroy(kamd_compact_lattice *c);
/* *reached_beam = 0 when determinization stopped early (memory limit) and the result is
 * pruned tighter than asked, the reference's "return false" case. */
int kamd_compact_lattice_sizes(const kamd_compact_lattice *c, int32_t *num_states, int32_t *num_arcs,
                               int32_t *num_labels, int32_t *start, int32_t *reached_beam);
int kamd_compact_lattice_get(const kamd_compact_lattice *c, float *state_final /* [2S] */,
                             int32_t *final_str_begin, int32_t *final_str_len /* [S] */,
                             kamd_clat_arc *arcs, int32_t *strings /* [num_labels] */);
/* CompactLatticeWriter entry (lat/kaldi-lattice.cc:62-94): "key " + the lattice, binary
 * ("compactlattice44" VectorFst) or text.  Acoustic costs are divided by acoustic_scale
 * (decoder/decoder-wrappers.cc:282-284) when it is neither 0 nor 1. */
int kamd_compact_lattice_write(const char *path, int append, const char *key, int binary,
                               const kamd_compact_lattice *c, float acoustic_scale);

int kamd_compact_lattice_scale_graph(kamd_compact_lattice *c, float scale);
/* fst::ScaleLattice with a diagonal scale (fstext/lattice-utils-inl.h:250-290): both weight
 * components of every arc and final weight. */
int kamd_compact_lattice_scale(kamd_compact_lattice *c, float graph_scale, float acoustic_scale);
/* A caller-owned copy (of a borrowed lattice, e.g. kamd_batch_decoder_get_compact_lattice's). */
kamd_compact_lattice *kamd_compact_lattice_copy(const kamd_compact_lattice *c);

/* ------------------------------------------------ const-ARPA LM rescoring -- */
/* ConstArpaLm (lm/const-arpa-lm.h:211-352): the compact n-gram LM the recipes rescore lattices
 * with (decode with tgsmall, then lattice-lmrescore-const-arpa with tglarge / fglarge: this is how
 * the reference's "tglarge" WERs are produced, egs/librispeech/s5/RESULTS).  The G.carpa on-disk
 * format is the reference's own (new format with tokens, and the old size-prefixed one). */
typedef struct kamd_const_arpa kamd_const_arpa;
/* arpa-to-const-arpa (BuildConstArpaLm, const-arpa-lm.cc:1064-1073) from an ARPA file of integer
 * word ids, or of words with a "word id" symbol table (words_txt; NULL = integers).  bos / eos
 * are required, unk = -1 when the LM has no <unk> (ArpaParseOptions). */
kamd_const_arpa *kamd_const_arpa_build(const char *arpa_path, int32_t bos, int32_t eos, int32_t unk,
                                       const char *words_txt);
kamd_const_arpa *kamd_const_arpa_read(const char *path);
int kamd_const_arpa_write(const kamd_const_arpa *lm, const char *path);
void kamd_const_arpa_destroy(kamd_const_arpa *lm);
int kamd_const_arpa_info(const kamd_const_arpa *lm, int32_t *bos, int32_t *eos, int32_t *unk, int32_t *order,
                         int32_t *num_words, int64_t *lm_states_size);
/* ConstArpaLm::GetNgramLogprob (const-arpa-lm.cc:741-779), natural log; FLT_MIN for a word the
 * LM cannot score (no <unk>). */
float kamd_const_arpa_ngram_logprob(const kamd_const_arpa *lm, int32_t word, const int32_t *hist, int n);
/* ArpaFileParser::Read alone (lm/arpa-file-parser.cc:41-262): the n-grams in file order with the
 * line they stand on, log-probabilities converted to natural log. */
int kamd_arpa_parse(const char *arpa_path, const char *words_txt, int32_t *counts, int counts_cap, int32_t *n_counts,
                    int32_t *lines, int32_t *orders, int32_t *words /* [cap][8] */, float *logprob, float *backoff,
                    int cap, int32_t *n);
/* lattice-lmrescore-const-arpa for one CompactLattice (latbin/lattice-lmrescore-const-arpa.cc:
 * 76-110): graph costs / lm_scale, ComposeCompactLatticeDeterministic with the LM
 * (lat/lattice-functions.cc:1529-1650), DeterminizeLattice, graph costs * lm_scale.  Arrays as
 * kamd_compact_lattice_get returns them.  NULL + "Empty lattice" when nothing composes. */
kamd_compact_lattice *kamd_compact_lattice_lmrescore_const_arpa(
    int32_t num_states, int32_t start, const float *state_final, const int32_t *final_str_begin,
    const int32_t *final_str_len, const kamd_clat_arc *arcs, int32_t num_arcs, const int32_t *strings,
    const kamd_const_arpa *lm, float lm_scale);

/* PruneActiveTokens in the middle of an utterance (decoder/lattice-faster-decoder.cc:519-546; the reference does it
 * every prune_interval frames): tokens and links the final backward sweep would drop anyway -- extra cost against the
 * current frontier above lattice_beam -- are dropped now and the survivors move to the bottom of the lane's arenas, so
 * a long utterance needs room for its PRUNED lattice plus the frames since the last compaction, not for everything
 * ever created.  Lanes must be un-finalized; decoding continues, the final lattice is the one an uncompacted decode
 * gives.  kamd_decoder_lane_usage (values of the last kamd_decoder_sync) tells when it is worth it. */
int kamd_decoder_compact(kamd_decoder *d, const int32_t *lanes, int n, void *stream);
int kamd_decoder_lane_usage(kamd_decoder *d, int lane, int32_t *tok_used, int32_t *tok_cap, int32_t *lnk_used,
                            int32_t *lnk_cap);
/* Best path of an UN-finalized lane (streaming partial results):
 * LatticeFasterOnlineDecoderTpl::GetBestPath = BestPathEnd + TraceBackBestPath
 * (decoder/lattice-faster-online-decoder.cc:54-165).  Requires a prior kamd_decoder_sync. */
int kamd_decoder_partial_best_path(kamd_decoder *d, int lane, int use_final_probs, int32_t *alignment,
                                   int ali_cap, int *ali_len, int32_t *words, int words_cap,
                                   int *words_len, float *graph_cost, float *acoustic_cost);
/* kamd_decoder_partial_best_path for n un-finalized lanes in one launch (a server's partial results after a tick).  alignments: [n][ali_cap],
 * words: [n][words_cap]; ali_len[i] = words_len[i] = -1 for a lane with no token alive.  Requires a prior kamd_decoder_sync. */
int kamd_decoder_partial_best_paths(kamd_decoder *d, const int32_t *lanes, int n, int use_final_probs, int32_t *alignments,
                                    int ali_cap, int32_t *ali_len, int32_t *words, int words_cap, int32_t *words_len,
                                    float *graph_cost, float *acoustic_cost);
/* The same with use_final_probs = 0 for a host that asks after EVERY tick: the decoder keeps each lane's previous answer
 * on the device and walks back only to the first frame whose token is the one recorded then (typically the newest few dozen
 * frames); identical results.  InitDecoding of a lane forgets its record. */
int kamd_decoder_partial_best_paths_incremental(kamd_decoder *d, const int32_t *lanes, int n, int32_t *alignments, int ali_cap,
                                                int32_t *ali_len, int32_t *words, int words_cap, int32_t *words_len,
                                                float *graph_cost, float *acoustic_cost);
/* What OnlineSilenceWeighting::ComputeCurrentTraceback (online2/online-ivector-feature.cc:464-510) reads off the
 * decoder, for n un-finalized lanes in one launch: the best path without final-probs, NEWEST frame first, one
 * (transition-id, token) pair per decoded frame; a token is named by its HCLG state (one token per state and
 * frame).  tids / tokens: [n][cap]; counts[i] = frames on the path, -1 when no token is alive.  Requires a prior
 * kamd_decoder_sync. */
int kamd_decoder_frame_tracebacks(kamd_decoder *d, const int32_t *lanes, int n, int32_t *tids, int32_t *tokens,
                                  int cap, int32_t *counts);
/* The same, incrementally: the decoder remembers (on the device, per lane) the token it reported for every frame, and
 * a walk stops at the first frame whose token is the remembered one -- the reference's own stopping rule
 * (online-ivector-feature.cc:489-496), so a tick costs the few frames whose best path changed, not the whole
 * utterance.  n_decoded[i] = NumFramesDecoded (-1: no token alive); n_entries[i] = pairs written for lane i, newest
 * frame first, the LAST one being the frame that matched (or frame 0).  kamd_decoder_init forgets a lane's record. */
int kamd_decoder_frame_tracebacks_incremental(kamd_decoder *d, const int32_t *lanes, int n, int32_t *tids,
                                              int32_t *tokens, int cap, int32_t *n_decoded, int32_t *n_entries);
/* ---- OnlineSilenceWeighting (online2/online-ivector-feature.h:404-535) + the delta-weight queue of
 * OnlineIvectorFeature (UpdateFrameWeights / UpdateStatsUntilFrameWeighted, .cc:159-174, 263-306), one object per
 * utterance.  tid_is_silence[tid] = 1 when TransitionIdToPhone(tid) is one of --silence-phones (index 0 unused);
 * max_state_duration <= 0 turns the duration rule off. */
typedef struct kamd_silence_weighting kamd_silence_weighting;
kamd_silence_weighting *kamd_silence_weighting_create(const uint8_t *tid_is_silence, int n_tids, float silence_weight,
                                                      float max_state_duration, int frame_subsampling_factor);
void kamd_silence_weighting_destroy(kamd_silence_weighting *w);
int kamd_silence_weighting_reset(kamd_silence_weighting *w);
/* ComputeCurrentTraceback: tids[k] / tokens[k] = frame num_frames_decoded - 1 - k (kamd_decoder_frame_tracebacks) */
int kamd_silence_weighting_compute_traceback(kamd_silence_weighting *w, int num_frames_decoded, const int32_t *tids,
                                             const int32_t *tokens, int n);
/* GetDeltaWeights(num_frames_ready_in) + UpdateFrameWeights: the (input frame, delta weight) pairs are queued */
int kamd_silence_weighting_get_delta_weights(kamd_silence_weighting *w, int num_frames_ready_in, int32_t *n_deltas);
/* What UpdateStatsUntilFrameWeighted(frame) hands to UpdateStatsForFrames: every queued delta for input frames
 * <= frame, duplicates summed, zero sums dropped, increasing frame order. */
int kamd_silence_weighting_pop_until(kamd_silence_weighting *w, int frame, int32_t *frames, float *weights, int cap,
                                     int32_t *n);
int kamd_silence_weighting_num_pending(const kamd_silence_weighting *w);
/* ---- endpointing (online2/online-endpoint.{h,cc}) ----
 * OnlineEndpointRule / OnlineEndpointConfig (online-endpoint.h:113-170); the silence phones are handed to the
 * decoder once (kamd_decoder_set_silence_phones) instead of travelling as a colon-separated string. */
typedef struct {
  int32_t must_contain_nonsilence;
  float min_trailing_silence, max_relative_cost, min_utterance_length;
} kamd_endpoint_rule;
typedef struct { kamd_endpoint_rule rule[5]; } kamd_endpoint_config;
void kamd_endpoint_config_default(kamd_endpoint_config *c);
/* EndpointDetected(config, num_frames_decoded, trailing_silence_frames, frame_shift_in_seconds,
 * final_relative_cost) (online-endpoint.cc:46-68): 1 / 0, -1 on a bad argument.  Host arithmetic only. */
int kamd_endpoint_detected(const kamd_endpoint_config *c, int num_frames_decoded, int trailing_silence_frames,
                           float frame_shift_in_seconds, float final_relative_cost);
/* tid2phone[1..num_tids] = TransitionModel::TransitionIdToPhone; silence_phones = --endpoint.silence-phones
 * (must be non-empty, no duplicates: online-endpoint.cc:77-82). */
int kamd_decoder_set_silence_phones(kamd_decoder *d, const int32_t *tid2phone, int32_t num_tids,
                                    const int32_t *silence_phones, int n_sil);
/* TrailingSilenceLength (online-endpoint.cc:71-102) of n un-finalized lanes in one launch: frames of silence
 * at the end of the best path taken WITHOUT final-probs.  Blocks. */
int kamd_decoder_trailing_silence_frames(kamd_decoder *d, const int32_t *lanes, int n, int32_t *out);
/* EndpointDetected(config, tmodel, frame_shift_in_seconds, decoder) (online-endpoint.cc:105-121) for n lanes;
 * frame_shift_in_seconds is that of the decoder's frames (feature shift x frame-subsampling-factor,
 * online-nnet3-decoding.cc:88-95).  detected[i] = 0 / 1; trailing_silence_frames may be NULL. */
int kamd_decoder_endpoint_detected(kamd_decoder *d, const kamd_endpoint_config *cfg, const int32_t *lanes, int n,
                                   float frame_shift_in_seconds, int32_t *detected, int32_t *trailing_silence_frames);
/* per-frame trace for parity debugging: ntok[f], cutoff[f], cost_offset[f]. */
int kamd_decoder_get_trace(kamd_decoder *d, int lane, int32_t *ntok,
                           float *cutoff, float *cost_offset, int cap);
/* Algorithmic work counters of SURVEY 8(d), accumulated since init of the lane:
 * [0]=N_exp tokens expanded, [1]=A_exp arcs iterated (emitting+epsilon),
 * [2]=A_emit emitting arcs, [3]=K_surv arcs passing the cutoff,
 * [4]=L_kept links written, [5]=N_tok tokens created, [6]=frames, [7]=reserved */
int kamd_decoder_get_counters(kamd_decoder *d, int lane, int64_t counters[8]);
/* Diagnostic: shader cycles thread 0 of the lane's workgroup spent in each phase since
 * init: [0] best-token reduce, [1] cutoff (count + radix select), [2] seed, [3] expand
 * (thread per token), [4] expand (wavefront / workgroup per hub), [5] epsilon closure,
 * [6] token compaction, [7] link fix-up, [8] epsilon links, [9] table clear + bookkeeping,
 * [10] finalize backward sweep, [11] finalize compaction. */
int kamd_decoder_get_phase_cycles(kamd_decoder *d, int lane, uint64_t cycles[16]);
/* device time (ms) of the AdvanceKernel launches since the last kamd_decoder_init (summed; at most
 * the last 8), measured with HIP events on the stream the kernel ran on (for bench.py's roofline),
 * and how many launches that was.  Valid after kamd_decoder_sync. */
float kamd_decoder_last_advance_ms(kamd_decoder *d);
int kamd_decoder_last_advance_launches(kamd_decoder *d);

/* -------------------------------------------------------------- pipeline -- */
/* wav -> features -> nnet -> decode for a batch of utterances, everything
 * device-resident between stages (the reference crosses the device boundary per
 * chunk: nnet-am-decodable-simple.cc:256,274; nnet-batch-compute.cc:412,469).
 * Mirrors nnet3-latgen-faster-batch's per-utterance flow
 * (nnet3bin/nnet3-latgen-faster-batch.cc:170-214). */
typedef struct kamd_pipeline kamd_pipeline;
kamd_pipeline *kamd_pipeline_create(kamd_feat *feat, kamd_nnet *nnet,
                                    kamd_decoder *dec);
void kamd_pipeline_destroy(kamd_pipeline *p);
/* Upload a batch of waveforms (host, concatenated; h_wave_off[n_utts+1]). */
int kamd_pipeline_load_batch(kamd_pipeline *p, const float *waves,
                             const int64_t *h_wave_off, int n_utts);
/* Features computed elsewhere (the features-rspecifier of nnet3bin/nnet3-latgen-faster.cc:166-200)
 * instead of waveforms: utterance u owns rows [row_off[u], row_off[u+1]) of host matrix `feats`
 * with `dim` columns; kamd_pipeline_run then skips the feature stage. */
int kamd_pipeline_load_features(kamd_pipeline *p, const float *feats, const int64_t *row_off, int n_utts, int dim);
/* One ivector per utterance of the resident batch ([n_utts x dim], host), the
 * `--ivectors` rspecifier of nnet3-latgen-faster (nnet3bin/nnet3-latgen-faster.cc:153-170).
 * dim = 0 clears them.  Online (per-chunk) ivectors are not supported: their values depend
 * on the reference's chunk boundaries (nnet-am-decodable-simple.cc:178-212). */
int kamd_pipeline_set_ivectors(kamd_pipeline *p, const float *ivectors, int dim);
/* --online-ivectors=... --online-ivector-period=N (nnet3bin/nnet3-latgen-faster.cc:60-75):
 * utterance u of the loaded batch owns rows [h_row_off[u], h_row_off[u+1]); the nnet stage then
 * runs chunk by chunk (kamd_nnet_forward_chunked_device).  dim <= 0 switches it off. */
int kamd_pipeline_set_online_ivectors(kamd_pipeline *p, const float *ivectors, const int64_t *h_row_off, int dim,
                                      int ivector_period, int frames_per_chunk);
/* Overlap of the nnet stage with the search inside one kamd_pipeline_run: the batch's output frames
 * are cut in time at n_bounds boundaries (output-frame indices, increasing); the log-likelihoods of
 * slice k+1 are computed on a second stream while the decoder lanes advance over slice k (the 64
 * lanes of the headline batch leave 192 CUs idle).  Lattices are identical to the unsliced run:
 * a frame's log-likelihood row does not depend on the slice it was computed in, and AdvanceDecoding
 * in pieces is AdvanceDecoding.  n_bounds = 0 (the default) turns it off.  Ignored (one slice) when
 * ivectors are set. */
int kamd_pipeline_set_overlap(kamd_pipeline *p, const int32_t *bounds, int n_bounds);
/* i-vectors estimated on the device from the batch's own features (instead of
 * kamd_pipeline_set_online_ivectors): run() then does features -> i-vectors -> chunked nnet
 * (frames_per_chunk as in DecodableNnetSimple) -> search.  e = NULL turns it off. */
int kamd_pipeline_set_ivector_extractor(kamd_pipeline *p, kamd_ivector_extractor *e, int frames_per_chunk);
/* Run the hot path over the resident batch: lanes 0..n_utts-1 hold the results.
 * Blocking; returns 0 or error.  stage_ms[3] = {features, nnet, decode(advance+
 * finalize)} device times from HIP events. */
int kamd_pipeline_run(kamd_pipeline *p, float stage_ms[4]);
/* device pointer + shape of the resident log-likelihoods of utterance u. */
int kamd_pipeline_get_loglikes(kamd_pipeline *p, int utt, float *out,
                               int rows_cap, int *rows, int *cols);
int kamd_pipeline_get_features(kamd_pipeline *p, int utt, float *out,
                               int rows_cap, int *rows, int *cols);

/* ------------------------------------------------ a test set per GPU ------ */
/* NnetBatchDecoder (nnet3/nnet-batch-compute.h:606-833; AcceptInput :665, Finished :690,
 * GetOutput :700) as used by nnet3-latgen-faster-batch (nnet3bin/nnet3-latgen-faster-batch.cc:
 * 170-214): all utterances of a shard resident in HBM; features in one launch, the acoustic
 * model in a few large passes, ONE work-queue launch of the decoder, and the per-utterance
 * host tail of DecodeUtteranceLatticeFaster (decoder/decoder-wrappers.cc:217-296: best path,
 * raw lattice, DeterminizeLatticePhonePrunedWrapper) on `host_threads` threads while the
 * search is still running. */
typedef struct {
  int32_t resident_lanes;     /* decoder lanes kept busy by the queue; 0 = one per compute unit */
  int32_t host_threads;       /* host-tail threads (NnetBatchDecoder's num_threads, :640) */
  int32_t determinize;        /* --determinize-lattice (decoder-wrappers.cc:262-277) */
  int32_t keep_raw_lattices;  /* keep the canonical raw lattice of every utterance on the host */
  int64_t nnet_pass_frames;   /* input frames per acoustic-model pass */
  int64_t lattice_pool_bytes; /* page-locked host pool the finished lattices wait in */
  float lattice_beam;         /* determinization beam = config.lattice_beam */
  kamd_determinize_opts det;
  int64_t first_pass_frames;  /* kamd_batch_decoder_load_host: input frames of the FIRST acoustic-model pass (small, so
                               * that the model starts behind a short upload); 0 = nnet_pass_frames */
} kamd_batch_opts;
void kamd_batch_opts_default(kamd_batch_opts *o);
typedef struct {
  float feat_ms, nnet_ms, decode_ms;   /* device stages (HIP events) */
  float host_tail_ms;                  /* wall time after the last utterance was published */
  float first_result_ms;               /* wall time until the first utterance was published */
  float total_ms;                      /* wall time of the whole run, results on the host */
  double nnet_flops;
  double host_thread_ms_sum;           /* CPU time spent in the host tail, all threads */
  int32_t lanes, nnet_passes, n_failed;
  int32_t long_utterances;             /* > 0: that many were searched beside the acoustic model (set_long_decoder) */
  /* kamd_batch_decoder_load_host only (0 otherwise): */
  float upload_ms;                     /* wall time until the last waveform byte was in HBM */
  float first_pass_start_ms;           /* wall time until the first pass's features were launched: the exposed part of the upload */
  float upload_wait_ms;                /* wall time the launching thread spent waiting for a pass's copies to be issued */
  int32_t upload_passes;
  float ivector_ms;                    /* device time of the online i-vector extraction (not part of nnet_ms), 0 without an extractor */
  int32_t n_retried;                   /* utterances searched a second time on a lane with larger arenas (their first search ran
                                        * out of token / link arena or lattice pool): inside decode_ms and total_ms */
  int32_t n_internal_events;           /* utterances whose FIRST search stopped on one of the lane's internal consistency checks
                                        * (kamd_queue_result.error & 32): not a capacity, a fault of the search itself.  They take
                                        * the second chance too, are counted here whether or not it is switched on, and are named
                                        * on stderr: anything but 0 is a defect to report */
} kamd_batch_stats;
typedef struct kamd_batch_decoder kamd_batch_decoder;
/* The stages are not owned.  tid_phone as kamd_lattice_determinize_phone_pruned (NULL: word
 * determinization only).  The decoder's max_lanes bounds resident_lanes; every lane has the
 * decoder's per-lane arena. */
kamd_batch_decoder *kamd_batch_decoder_create(kamd_feat *feat, kamd_nnet *nnet, kamd_decoder *dec,
                                              const kamd_batch_opts *opts, const int32_t *tid_phone,
                                              int32_t num_tids);
void kamd_batch_decoder_destroy(kamd_batch_decoder *b);
/* Optional, for shards so small that the search is bound by the longest utterance's own chain of
 * frames (a rank of an 8-GPU run over test-clean holds ~330 utterances for 256 lanes): `dec_long`
 * is a second decoder object over the same graph with `lanes` lanes (each sized for the longest
 * utterance).  run() then scores the `lanes` longest utterances first and searches them on
 * `dec_long` and a second stream while the acoustic model of the others is still running -- when
 * 0.55 x the longest utterance's frames exceed 1.5 x the shard's frames per lane; otherwise, and
 * for models with an i-vector input, nothing changes.  Results are the same either way (the reference's
 * NnetBatchDecoder likewise leaves the order of computation open, nnet-batch-compute.h:606-660).
 * NULL / 0 lanes switches it off.  The caller keeps ownership of dec_long. */
int kamd_batch_decoder_set_long_decoder(kamd_batch_decoder *b, kamd_decoder *dec_long, int lanes);
/* AcceptInput for the whole shard: utterance u owns samples [wave_off[u], wave_off[u+1]). */
int kamd_batch_decoder_load(kamd_batch_decoder *b, const float *waves, const int64_t *wave_off, int n_utts);
/* AcceptInput with the waveforms left where they are, in the caller's host memory: every run() uploads them itself,
 * inside what it times -- pass by pass on a copy stream, the copies of pass k + 1 overlapped with the features and the
 * acoustic model of pass k; the first pass is small (kamd_batch_opts.first_pass_frames).  This call page-locks the
 * buffer in place (hipHostRegister, undone by the next load / destroy); where the runtime refuses, run() stages the
 * samples through four 32 MB page-locked buffers instead.  `waves` must stay valid until the next load or destroy.  (nnet3-latgen-faster-
 * batch reads each wave / feature matrix from its table inside the timed loop too: nnet3-latgen-faster-batch.cc:176-214.) */
int kamd_batch_decoder_load_host(kamd_batch_decoder *b, const float *waves, const int64_t *wave_off, int n_utts);
/* Releases the buffer of kamd_batch_decoder_load_host: undoes the page lock and forgets the pointer, so that the caller may
 * free `waves` while the decoder object lives on (outputs of the last run() stay readable; another run() needs a new load).
 * A no-op when nothing is held. */
int kamd_batch_decoder_unload_host(kamd_batch_decoder *b);
/* Workload synthesis for benchmarks: the search reads log-likelihoods from d_loglikes (device, [total output frames x
 * P], utterances back to back in load order, as kamd_batch_decoder_get_loglikes numbers them) instead of the acoustic
 * model's output, which is still computed.  NULL switches it off. */
int kamd_batch_decoder_set_loglike_override(kamd_batch_decoder *b, const float *d_loglikes);
/* Benchmark workload synthesis (kaldi_amd/csrc/synth.hip; not part of the decode path): d_out[r][p] = noise * N(0, 1)
 * + (p == d_true_pdf[r] ? peak : 0), the planted-transcript log-likelihoods of kaldi_amd/synth.py on the device. */
int kamd_synth_planted_loglikes_device(float *d_out, int64_t rows, int num_pdfs, int ld, const int32_t *d_true_pdf, float peak,
                                       float noise, uint64_t seed, void *stream);
/* Output frames of every loaded utterance (0 for the ones too short for a frame); total returned. */
int64_t kamd_batch_decoder_output_frames(kamd_batch_decoder *b, int32_t *frames, int cap);
/* The recipe's online i-vectors (steps/nnet3/decode.sh:105-107 passes --online-ivectors=scp:... --online-ivector-period=N,
 * matrices that steps/online/nnet2/extract_ivectors_online.sh wrote with ivector-extract-online2): estimated HERE from the
 * features of every pass (kamd_ivector_extract_online_device), and the acoustic model then evaluates chunk by chunk like
 * DecodableNnetSimple (nnet3/nnet-am-decodable-simple.cc:93-214: frames_per_chunk input frames a chunk, its own context
 * recomputed, the i-vector row GetCurrentIvector picks; kamd_nnet_forward_chunked_device).  Set before the set is loaded
 * (kamd_batch_decoder_load / _load_host); NULL removes it.  The extractor is not owned. */
int kamd_batch_decoder_set_ivector_extractor(kamd_batch_decoder *b, kamd_ivector_extractor *e, int frames_per_chunk);
/* Which of the reference's two chunkings the acoustic model of a set with online i-vectors follows: 0 (default)
 * DecodableNnetSimple's (nnet3-latgen-faster: kamd_nnet_forward_chunked_device), 1 NnetBatchComputer's tasks
 * (nnet3-latgen-faster-batch: kamd_nnet_forward_tasks_device).  They differ in where chunks start and therefore in the
 * i-vector row a frame is evaluated with. */
int kamd_batch_decoder_set_chunk_rule(kamd_batch_decoder *b, int rule);
/* AcceptInput as the reference declares it (nnet-batch-compute.h:665-669: feature matrices, not
 * waveforms, plus the optional per-utterance i-vector): rows [row_off[u], row_off[u+1]) of feats
 * (dim floats per row, dim = the model's input dim) are utterance u; ivectors is [n_utts x
 * ivector_dim] or NULL.  The run then starts at the acoustic model (feat may be NULL at create). */
int kamd_batch_decoder_load_features(kamd_batch_decoder *b, const float *feats, const int64_t *row_off, int dim,
                                     const float *ivectors, int ivector_dim, int n_utts);
/* Finished(): runs the shard and returns when every utterance's output is on the host.  A
 * per-utterance failure (capacity) does not fail the run: see n_failed and _get_output. */
int kamd_batch_decoder_run(kamd_batch_decoder *b, kamd_batch_stats *stats);
/* GetOutput: best path (words, transition-ids, total weight) and the device record. */
int kamd_batch_decoder_get_output(kamd_batch_decoder *b, int utt, int32_t *words, int words_cap, int *words_len,
                                  int32_t *alignment, int ali_cap, int *ali_len, float *graph_cost,
                                  float *acoustic_cost, kamd_queue_result *record);
/* Borrowed pointers, valid until the next load / run / destroy. */
int kamd_batch_decoder_get_raw_lattice(kamd_batch_decoder *b, int utt, int32_t *num_states, int32_t *num_arcs,
                                       int32_t *start, const int32_t **state_frame, const int32_t **state_hclg,
                                       const float **state_cost, const float **state_final,
                                       const kamd_lat_arc **arcs);
const kamd_compact_lattice *kamd_batch_decoder_get_compact_lattice(kamd_batch_decoder *b, int utt);
int kamd_batch_decoder_get_loglikes(kamd_batch_decoder *b, int utt, float *out, int rows_cap, int *rows, int *cols);

#ifdef __cplusplus
}
#endif
#endif /* KALDI_AMD_H_ */
