// online.hip -- streaming wrappers (BASELINE config 5: online2-wav-nnet3-latgen-faster).
//
//  * kamd_online_feat: OnlineGenericBaseFeature<C> (feat/online-feature.cc:63-200):
//    AcceptWaveform / InputFinished / NumFramesReady / GetFrames.  The waveform seen so
//    far stays in HBM; newly completed frames are computed by the same FeatKernel as the
//    offline path (online == offline is what feat/online-feature-test.cc asserts).
//  * kamd_nnet_forward_range: the rows DecodableAmNnetLoopedOnline would serve
//    (nnet3/decodable-online-looped.cc:56-240): output frames whose right context is
//    available, or all of them once the input is finished (edge frames clamped).
#include <algorithm>
#include <vector>

#include "common.h"

namespace kamd { float *NnetScratch(kamd_nnet *h, size_t floats); }   // nnet.hip

extern "C" int kamd_feat_compute_frames_device(kamd_feat *f, const float *d_wave, int64_t num_samples,
                                               int first_frame, int num_frames, float *d_out, int ld_out,
                                               void *stream);
extern "C" int kamd_feat_num_frames_flush(const kamd_feat *f, int64_t num_samples, int flush);

namespace kamd {
struct OnlineFeat {
  kamd_feat *feat;
  float *d_wave = NULL; size_t wave_cap = 0; int64_t n_samp = 0;
  float *d_frames = NULL; size_t frames_cap = 0; int n_frames = 0;   // [n_frames x ld]
  int dim = 0, ld = 0;
  bool finished = false;
};
}  // namespace kamd
using kamd::OnlineFeat;

extern "C" {

kamd_online_feat *kamd_online_feat_create(kamd_feat *feat) {
  if (!kamd::RequireDevice()) return NULL;
  OnlineFeat *o = new OnlineFeat();
  o->feat = feat;
  o->dim = kamd_feat_dim(feat);
  o->ld = kamd::RoundUp(o->dim, 16);
  return reinterpret_cast<kamd_online_feat *>(o);
}
void kamd_online_feat_destroy(kamd_online_feat *h) {
  OnlineFeat *o = reinterpret_cast<OnlineFeat *>(h);
  if (!o) return;
  if (o->d_wave) (void)hipFree(o->d_wave);
  if (o->d_frames) (void)hipFree(o->d_frames);
  delete o;
}

static int ComputeNew(OnlineFeat *o) {
  // OnlineGenericBaseFeature::ComputeFeatures (feat/online-feature.cc:150-190)
  const int n_new = kamd_feat_num_frames_flush(o->feat, o->n_samp, o->finished ? 1 : 0);
  if (n_new <= o->n_frames) return KAMD_OK;
  const size_t need = static_cast<size_t>(n_new) * o->ld;
  if (need > o->frames_cap) {
    float *p = NULL;
    const size_t cap = std::max(need * 2, static_cast<size_t>(1024) * o->ld);
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&p), cap * sizeof(float)));
    if (o->d_frames) {
      KAMD_HIP(hipMemcpy(p, o->d_frames, static_cast<size_t>(o->n_frames) * o->ld * sizeof(float), hipMemcpyDeviceToDevice));
      KAMD_HIP(hipFree(o->d_frames));
    }
    o->d_frames = p; o->frames_cap = cap;
  }
  int rc = kamd_feat_compute_frames_device(o->feat, o->d_wave, o->n_samp, o->n_frames, n_new - o->n_frames,
                                           o->d_frames + static_cast<size_t>(o->n_frames) * o->ld, o->ld, NULL);
  if (rc != KAMD_OK) return rc;
  o->n_frames = n_new;
  return KAMD_OK;
}

int kamd_online_feat_accept_waveform(kamd_online_feat *h, float sampling_rate, const float *wave, int64_t n) {
  OnlineFeat *o = reinterpret_cast<OnlineFeat *>(h);
  (void)sampling_rate;   // must equal --sample-frequency; resampling is out of scope
  if (o->finished) return kamd::SetError(KAMD_ERR_STATE, "AcceptWaveform called after InputFinished");  // online-feature.cc:126-127
  if (n <= 0) return KAMD_OK;
  const size_t need = static_cast<size_t>(o->n_samp + n);
  if (need > o->wave_cap) {
    float *p = NULL;
    const size_t cap = std::max(need * 2, static_cast<size_t>(1) << 16);
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&p), cap * sizeof(float)));
    if (o->d_wave) {
      KAMD_HIP(hipMemcpy(p, o->d_wave, static_cast<size_t>(o->n_samp) * sizeof(float), hipMemcpyDeviceToDevice));
      KAMD_HIP(hipFree(o->d_wave));
    }
    o->d_wave = p; o->wave_cap = cap;
  }
  KAMD_HIP(hipMemcpy(o->d_wave + o->n_samp, wave, static_cast<size_t>(n) * sizeof(float), hipMemcpyHostToDevice));
  o->n_samp += n;
  return ComputeNew(o);
}
int kamd_online_feat_input_finished(kamd_online_feat *h) {
  OnlineFeat *o = reinterpret_cast<OnlineFeat *>(h);
  o->finished = true;
  return ComputeNew(o);
}
int kamd_online_feat_num_frames_ready(const kamd_online_feat *h) { return reinterpret_cast<const OnlineFeat *>(h)->n_frames; }
int kamd_online_feat_is_last_frame(const kamd_online_feat *h, int frame) {
  const OnlineFeat *o = reinterpret_cast<const OnlineFeat *>(h);
  return o->finished && frame == o->n_frames - 1;   // online-feature.h: IsLastFrame
}
int kamd_online_feat_get_frames(kamd_online_feat *h, int first, int n, float *out) {
  OnlineFeat *o = reinterpret_cast<OnlineFeat *>(h);
  if (first < 0 || n < 0 || first + n > o->n_frames) return kamd::SetError(KAMD_ERR_ARG, "frames [%d,%d) not ready (%d)", first, first + n, o->n_frames);
  if (n == 0) return KAMD_OK;
  KAMD_HIP(hipMemcpy2D(out, o->dim * sizeof(float), o->d_frames + static_cast<size_t>(first) * o->ld, o->ld * sizeof(float),
                       o->dim * sizeof(float), n, hipMemcpyDeviceToHost));
  return KAMD_OK;
}
const float *kamd_online_feat_device_frames(const kamd_online_feat *h, int *ld) {
  const OnlineFeat *o = reinterpret_cast<const OnlineFeat *>(h);
  *ld = o->ld;
  return o->d_frames;
}

int kamd_nnet_num_frames_ready(const kamd_nnet *n, int feat_frames_ready, int input_finished) {
  // DecodableAmNnetLoopedOnline::NumFramesReady (nnet3/decodable-online-looped.cc:56-85),
  // without the whole-chunk rounding: an output frame is ready once its right context is.
  const int R = kamd_nnet_right_context(n);
  if (feat_frames_ready <= 0) return 0;
  const int total = kamd_nnet_num_output_frames(n, feat_frames_ready);
  if (input_finished) return total;
  // largest o with s*o + R <= ready-1
  const int s = kamd_nnet_frame_subsampling_factor(n);
  const int last = feat_frames_ready - 1 - R;
  return last < 0 ? 0 : std::min(total, last / s + 1);
}

int kamd_nnet_forward_range(kamd_nnet *n, const float *d_feats, int ld_in, int feat_frames_ready,
                            int input_finished, int out_first, int out_count, float *d_out, int ld_out) {
  if (out_count <= 0) return KAMD_OK;
  const int s = kamd_nnet_frame_subsampling_factor(n);
  const int L = kamd_nnet_left_context(n), R = kamd_nnet_right_context(n);
  if (out_first < 0 || out_first + out_count > kamd_nnet_num_frames_ready(n, feat_frames_ready, input_finished))
    return kamd::SetError(KAMD_ERR_ARG, "output frames [%d,%d) are not ready", out_first, out_first + out_count);
  // slice of the input that holds every needed frame, starting on the subsampling grid
  const int k0 = std::min(out_first, (L + s - 1) / s);          // discarded leading outputs
  const int in_first = s * (out_first - k0);
  const int in_last = std::min(feat_frames_ready - 1, s * (out_first + out_count - 1) + R);
  const int T = in_last - in_first + 1;
  const int n_out = kamd_nnet_num_output_frames(n, T);
  const int P = kamd_nnet_output_dim(n);
  // (the model's own grow-only scratch: a stream calls this after every chunk, and hipMalloc / hipFree wait for the device)
  float *d_tmp = kamd::NnetScratch(n, static_cast<size_t>(n_out) * P);
  if (!d_tmp) return kamd::SetError(KAMD_ERR_HIP, "kamd_nnet_forward_range: out of device memory");
  int64_t in_off[2] = {0, T}, out_off[1] = {0};
  int rc = kamd_nnet_forward_batch_device(n, d_feats + static_cast<size_t>(in_first) * ld_in, in_off, ld_in, NULL, 1,
                                          d_tmp, out_off, P, NULL);
  if (rc == KAMD_OK) {
    hipError_t e = hipMemcpy2DAsync(d_out, ld_out * sizeof(float), d_tmp + static_cast<size_t>(k0) * P, P * sizeof(float),
                                    P * sizeof(float), out_count, hipMemcpyDeviceToDevice, NULL);
    if (e == hipSuccess) e = hipStreamSynchronize(NULL);      // (as before: the rows are there when the call returns)
    if (e != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "copy failed: %s", hipGetErrorString(e));
  }
  return rc;
}

// ------------------------------------------------------------------ batched streaming
// N concurrent streams (BASELINE configs[4] at scale): every stream is a decoder lane with a
// slot in one pooled waveform buffer and one pooled feature buffer.  One tick =
//   * ONE feature launch for the frames that became computable on all ticking streams,
//   * ONE batched nnet forward over their new slices (each slice carries its own left / right
//     context; slices of different streams are separate items of the same GEMMs),
//   * ONE AdvanceKernel launch (a lane per stream, the new rows of each).
// What a single-stream SingleUtteranceNnet3Decoder does per chunk (online2/online-nnet3-decoding.cc:
// 40-97), for all streams at once; the rows served are those of kamd_nnet_forward_range, i.e.
// bit-equal to the offline forward (tests/test_gpu_online.py).
extern "C" int kamd_feat_compute_ranges_device(kamd_feat *h, const float *d_waves, const int64_t *h_wave_start,
                                               const int64_t *h_wave_len, const int32_t *h_first_frame,
                                               const int32_t *h_num_frames, int n, float *d_out,
                                               const int64_t *h_row_off, int ld_out, void *stream);
extern "C" int kamd_nnet_forward_slices_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_start,
                                               const int32_t *h_in_len, int ld_in, const float *d_ivectors, int n_items,
                                               float *d_out, const int64_t *h_out_row_off, int ld_out, void *stream);
extern "C" int kamd_nnet_forward_slices_slots_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_start, const int32_t *h_in_len,
                                                     int ld_in, const float *d_ivector_table, int table_rows, int period,
                                                     const int32_t *h_slot_base, const int32_t *h_slot_first, const int32_t *h_slot_count,
                                                     const int32_t *h_abs_t0, int n_items, float *d_out, const int64_t *h_out_row_off,
                                                     int ld_out, void *stream);
namespace kamd {
// rows src[src_row[i]] -> dst[dst_row[i] .. dst_row[i] + count[i]) (a stream's new i-vector slots all get the tick's estimate)
__global__ void AssignSlotsKernel(const float *src, float *dst, const int *src_row, const int *dst_row, const int *count, int dim) {
  const int i = blockIdx.x;
  for (int r = 0; r < count[i]; r++)
    for (int k = threadIdx.x; k < dim; k += blockDim.x) dst[static_cast<size_t>(dst_row[i] + r) * dim + k] = src[static_cast<size_t>(src_row[i]) * dim + k];
}
struct StreamBatch {
  kamd_feat *feat; kamd_nnet *nnet; kamd_decoder *dec;
  // online i-vectors (kamd_stream_batch_set_ivector_extractor): DecodableNnetLoopedOnline's schedule.  The
  // network is served chunk by chunk of `chunk` input frames (chunk k once (k+1)*chunk + right context frames
  // are there, or the input is finished); when a tick makes new chunks computable the stream's estimate is
  // advanced ONCE to the most recent frame (OnlineIvectorFeature::GetFrame, use_most_recent_ivector) and every
  // i-vector slot floor(t / chunk) those chunks' input ranges introduce gets it (nnet-compile-looped.cc:186-207).
  kamd_ivector_extractor *ie = NULL;
  kamd_ivector_workspace *ie_ws = NULL;   // this batch's LDA / posterior rows (the weighted update re-reads earlier frames' rows)
  int chunk = 0, iv_dim = 0, splice_right = 0, slot_first = 0, max_slots = 0, rec_size = 0;
  double *d_rec = NULL;          // [S][rec_size]
  float *d_slots = NULL;         // [S][max_slots][iv_dim]
  float *d_est = NULL;           // [S][iv_dim]: the tick's estimates (row = position in the update call)
  int *d_assign = NULL;          // 3 x S ints
  std::vector<int> chunks_done, iv_done, slots_assigned;
  // OnlineSilenceWeighting, one per stream (kamd_stream_batch_set_silence_weighting); empty = off
  std::vector<kamd_silence_weighting *> sw;
  // PruneActiveTokens as arena compaction: a stream whose token or link arena is fuller than this fraction at the start
  // of a tick is compacted first (kamd_decoder_compact); 0 = never
  float compact_at = 0.5f;
  // LatticeFasterDecoderConfig::prune_interval for streams (lattice-faster-decoder.cc:617-619: PruneActiveTokens every
  // prune_interval frames): a stream that has decoded this many frames since its last compaction is compacted at the start of
  // its next tick.  0 = never (FinalizeDecoding then sweeps every frame of the utterance at once: the end-of-utterance latency).
  int prune_interval = 0;
  std::vector<int> compacted_at;        // frames decoded at the stream's last compaction
  int64_t n_compactions = 0;
  int S = 0, dim = 0, ld = 0, P = 0, max_frames = 0;
  int64_t max_samples = 0;
  float *d_wave = NULL, *d_frames = NULL, *d_ll = NULL;
  void *d_stage = NULL; size_t stage_cap = 0;       // accept_many: {segment table, concatenated chunks}
  size_t ll_cap = 0;
  std::vector<int64_t> n_samp;
  std::vector<int> n_frames, decoded;
  std::vector<char> finished, live;
  std::vector<int> status;     // per stream: 0 = fine, else the decoder's capacity flags: the stream is out until restarted
};
}  // namespace kamd
using kamd::StreamBatch;

kamd_stream_batch *kamd_stream_batch_create(kamd_feat *feat, kamd_nnet *nnet, kamd_decoder *dec, int max_streams,
                                            float max_seconds, float samp_freq) {
  if (!kamd::RequireDevice()) return NULL;
  if (max_streams <= 0 || max_seconds <= 0) { kamd::SetError(KAMD_ERR_ARG, "bad stream batch size"); return NULL; }
  StreamBatch *b = new StreamBatch();
  b->feat = feat; b->nnet = nnet; b->dec = dec; b->S = max_streams;
  b->dim = kamd_feat_dim(feat); b->ld = kamd::RoundUp(b->dim, 16); b->P = kamd_nnet_output_dim(nnet);
  b->max_samples = static_cast<int64_t>(max_seconds * samp_freq) + 1;
  b->max_frames = kamd_feat_num_frames_flush(feat, b->max_samples, 1) + 1;
  b->n_samp.assign(max_streams, 0); b->n_frames.assign(max_streams, 0); b->decoded.assign(max_streams, 0);
  b->finished.assign(max_streams, 0); b->live.assign(max_streams, 0); b->status.assign(max_streams, 0);
  if (hipMalloc(reinterpret_cast<void **>(&b->d_wave), static_cast<size_t>(max_streams) * b->max_samples * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&b->d_frames), static_cast<size_t>(max_streams) * b->max_frames * b->ld * sizeof(float)) != hipSuccess ||
      hipMemset(b->d_frames, 0, static_cast<size_t>(max_streams) * b->max_frames * b->ld * sizeof(float)) != hipSuccess) {
    kamd::SetError(KAMD_ERR_HIP, "stream batch allocation failed");
    if (b->d_wave) (void)hipFree(b->d_wave);
    if (b->d_frames) (void)hipFree(b->d_frames);
    delete b;
    return NULL;
  }
  return reinterpret_cast<kamd_stream_batch *>(b);
}

void kamd_stream_batch_destroy(kamd_stream_batch *h) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (!b) return;
  if (b->d_wave) (void)hipFree(b->d_wave);
  if (b->d_frames) (void)hipFree(b->d_frames);
  if (b->d_ll) (void)hipFree(b->d_ll);
  if (b->d_stage) (void)hipFree(b->d_stage);
  if (b->d_rec) (void)hipFree(b->d_rec);
  if (b->d_slots) (void)hipFree(b->d_slots);
  if (b->d_est) (void)hipFree(b->d_est);
  if (b->d_assign) (void)hipFree(b->d_assign);
  for (kamd_silence_weighting *w : b->sw) kamd_silence_weighting_destroy(w);
  if (b->ie_ws) kamd_ivector_workspace_destroy(b->ie_ws);
  delete b;
}

int kamd_stream_batch_set_ivector_extractor(kamd_stream_batch *h, kamd_ivector_extractor *ie, int frames_per_chunk, int splice_right) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  for (int s = 0; s < b->S; s++) if (b->live[s]) return kamd::SetError(KAMD_ERR_STATE, "set the i-vector extractor before any stream is started");
  const int sub = kamd_nnet_frame_subsampling_factor(b->nnet);
  if (!ie || frames_per_chunk <= 0 || splice_right < 0) return kamd::SetError(KAMD_ERR_ARG, "bad i-vector extractor / frames per chunk");
  // GetChunkSize (nnet3/nnet-utils / decodable-simple-looped.cc:52-56): rounded up to a multiple of the subsampling factor
  frames_per_chunk = sub * ((frames_per_chunk + sub - 1) / sub);
  b->ie = ie; b->chunk = frames_per_chunk; b->iv_dim = kamd_ivector_dim(ie); b->splice_right = splice_right;
  const int lc = kamd_nnet_left_context(b->nnet), rc = kamd_nnet_right_context(b->nnet);
  b->slot_first = -((lc + frames_per_chunk - 1) / frames_per_chunk);                   // floor(-lc / chunk)
  b->max_slots = (b->max_frames + rc + frames_per_chunk) / frames_per_chunk + 2 - b->slot_first;
  b->rec_size = kamd_ivector_stream_record_size(ie);
  b->chunks_done.assign(b->S, 0); b->iv_done.assign(b->S, 0); b->slots_assigned.assign(b->S, 0);
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_rec), static_cast<size_t>(b->S) * b->rec_size * sizeof(double)));
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_slots), static_cast<size_t>(b->S) * b->max_slots * b->iv_dim * sizeof(float)));
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_est), static_cast<size_t>(b->S + 1) * b->iv_dim * sizeof(float)));
  KAMD_HIP(hipMemset(b->d_est, 0, static_cast<size_t>(b->S + 1) * b->iv_dim * sizeof(float)));     // row S stays zero
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_assign), static_cast<size_t>(3) * b->S * sizeof(int)));
  return KAMD_OK;
}

// like kamd_stream_batch_start, each stream with its speaker's adaptation state (states: n records of
// kamd_ivector_state_size() doubles, or NULL = fresh): SetAdaptationState (online2-wav-nnet3-latgen-faster.cc:232-236)
int kamd_stream_batch_start_adapted(kamd_stream_batch *h, const int32_t *streams, int n, const double *states) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (!b->ie) return kamd::SetError(KAMD_ERR_STATE, "no i-vector extractor set");
  const int rc = kamd_stream_batch_start(h, streams, n);      // fresh records
  if (rc != KAMD_OK || !states) return rc;
  const int SS = kamd_ivector_state_size(b->ie);
  std::vector<double> rec(b->rec_size);
  for (int i = 0; i < n; i++) {
    kamd_ivector_stream_record_init(b->ie, states + static_cast<size_t>(i) * SS, rec.data());
    KAMD_HIP(hipMemcpy(b->d_rec + static_cast<size_t>(streams[i]) * b->rec_size, rec.data(), b->rec_size * sizeof(double), hipMemcpyHostToDevice));
  }
  return KAMD_OK;
}

// diagnostic / tests: the i-vector slots assigned so far (slot first_slot + j = time range [(first_slot + j) * chunk, +chunk))
int kamd_stream_batch_get_ivector_slots(kamd_stream_batch *h, int stream, float *out, int rows_cap, int *first_slot, int *count) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (!b->ie) return kamd::SetError(KAMD_ERR_STATE, "no i-vector extractor set");
  if (stream < 0 || stream >= b->S) return kamd::SetError(KAMD_ERR_ARG, "stream %d out of range", stream);
  *first_slot = b->slot_first; *count = b->slots_assigned[stream];
  if (*count > rows_cap) return kamd::SetError(KAMD_ERR_ARG, "buffer too small for %d slots", *count);
  KAMD_HIP(hipDeviceSynchronize());
  if (*count > 0)
    KAMD_HIP(hipMemcpy(out, b->d_slots + static_cast<size_t>(stream) * b->max_slots * b->iv_dim, static_cast<size_t>(*count) * b->iv_dim * sizeof(float),
                       hipMemcpyDeviceToHost));
  return KAMD_OK;
}

// --ivector-silence-weighting.* (online2-wav-nnet3-latgen-faster.cc:214-216, 258-266): before every AdvanceDecoding
// the streams' current tracebacks re-decide which frames count as silence, and the i-vector statistics are
// corrected by the difference.  silence_weight == 1 (or no silence transition-id) is the reference's "not Active".
int kamd_stream_batch_set_silence_weighting(kamd_stream_batch *h, const uint8_t *tid_is_silence, int n_tids, float silence_weight,
                                            float max_state_duration) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  for (int s = 0; s < b->S; s++) if (b->live[s]) return kamd::SetError(KAMD_ERR_STATE, "set the silence weighting before any stream is started");
  if (!b->ie) return kamd::SetError(KAMD_ERR_STATE, "silence weighting acts on the i-vector statistics: set the i-vector extractor first");
  for (kamd_silence_weighting *w : b->sw) kamd_silence_weighting_destroy(w);
  b->sw.clear();
  bool any = false;
  for (int t = 0; tid_is_silence && t < n_tids; t++) any = any || tid_is_silence[t];
  if (!any || silence_weight == 1.0f) return KAMD_OK;
  for (int s = 0; s < b->S; s++) {
    kamd_silence_weighting *w = kamd_silence_weighting_create(tid_is_silence, n_tids, silence_weight, max_state_duration,
                                                              kamd_nnet_frame_subsampling_factor(b->nnet));
    if (!w) return KAMD_ERR_ARG;
    b->sw.push_back(w);
  }
  return KAMD_OK;
}

int kamd_stream_batch_set_compaction(kamd_stream_batch *h, float fraction) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (!(fraction >= 0.0f && fraction < 1.0f)) return kamd::SetError(KAMD_ERR_ARG, "compaction threshold must be in [0, 1)");
  b->compact_at = fraction;
  return KAMD_OK;
}
int kamd_stream_batch_set_prune_interval(kamd_stream_batch *h, int frames) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (frames < 0) return kamd::SetError(KAMD_ERR_ARG, "prune interval must be >= 0");
  b->prune_interval = frames;
  return KAMD_OK;
}
int64_t kamd_stream_batch_num_compactions(const kamd_stream_batch *h) { return reinterpret_cast<const StreamBatch *>(h)->n_compactions; }

// GetAdaptationState after the utterance (online2-wav-nnet3-latgen-faster.cc:284): the i-vector statistics as they
// are plus the speaker CMVN statistics advanced by every frame of this utterance; LimitFrames is the caller's
// (kamd_ivector_state_limit_frames)
int kamd_stream_batch_get_adaptation_state(kamd_stream_batch *h, int stream, double *state) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (!b->ie) return kamd::SetError(KAMD_ERR_STATE, "no i-vector extractor set");
  if (stream < 0 || stream >= b->S || !b->live[stream]) return kamd::SetError(KAMD_ERR_ARG, "stream %d is not started", stream);
  const int SS = kamd_ivector_state_size(b->ie);
  KAMD_HIP(hipDeviceSynchronize());
  KAMD_HIP(hipMemcpy(state, b->d_rec + static_cast<size_t>(stream) * b->rec_size, SS * sizeof(double), hipMemcpyDeviceToHost));
  if (b->n_frames[stream] > 0) {
    const int64_t off[2] = {static_cast<int64_t>(stream) * b->max_frames, static_cast<int64_t>(stream) * b->max_frames + b->n_frames[stream]};
    return kamd_cmvn_acc_stats_device(b->d_frames, off, b->ld, b->dim, 1, state, NULL);
  }
  return KAMD_OK;
}

// a new utterance on these streams: SingleUtteranceNnet3DecoderTpl's constructor calls
// decoder_.InitDecoding() (online-nnet3-decoding.cc:40)
int kamd_stream_batch_start(kamd_stream_batch *h, const int32_t *streams, int n) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  for (int i = 0; i < n; i++)
    if (streams[i] < 0 || streams[i] >= b->S) return kamd::SetError(KAMD_ERR_ARG, "stream %d out of range", streams[i]);
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    b->n_samp[s] = 0; b->n_frames[s] = 0; b->decoded[s] = 0; b->finished[s] = 0; b->status[s] = 0;
    if (b->ie) {                  // a fresh adaptation state; kamd_stream_batch_start_adapted replaces it
      std::vector<double> rec(b->rec_size);
      kamd_ivector_stream_record_init(b->ie, NULL, rec.data());
      KAMD_HIP(hipMemcpy(b->d_rec + static_cast<size_t>(s) * b->rec_size, rec.data(), b->rec_size * sizeof(double), hipMemcpyHostToDevice));
      b->chunks_done[s] = 0; b->iv_done[s] = 0; b->slots_assigned[s] = 0;
    }
    if (!b->sw.empty()) kamd_silence_weighting_reset(b->sw[s]);    // "initialize a new copy of this object for each new utterance"
    b->live[s] = 1;
  }
  int rc = kamd_decoder_init(b->dec, streams, n, NULL);
  if (rc != KAMD_OK) return rc;
  // only these lanes matter here: another stream's earlier overflow is that stream's business
  std::vector<int32_t> err(std::max(n, 1), 0);
  rc = kamd_decoder_sync_lanes(b->dec, streams, n, err.data());
  if (rc != KAMD_OK) return rc;
  for (int i = 0; i < n; i++)
    if (err[i]) { b->status[streams[i]] = err[i]; return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: InitDecoding overflowed the lane (flags %d)", streams[i], err[i]); }
  return KAMD_OK;
}

// AcceptWaveform / InputFinished (online-nnet3-decoding.h:77-86): samples are appended to the
// stream's slot in HBM; nothing is computed until the next tick.
int kamd_stream_batch_accept(kamd_stream_batch *h, int stream, const float *wave, int64_t n, int input_finished) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (stream < 0 || stream >= b->S || !b->live[stream]) return kamd::SetError(KAMD_ERR_ARG, "stream %d is not started", stream);
  if (b->finished[stream] && n > 0) return kamd::SetError(KAMD_ERR_STATE, "AcceptWaveform called after InputFinished");
  if (b->n_samp[stream] + n > b->max_samples) return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: more audio than max_seconds", stream);
  if (n > 0) {
    KAMD_HIP(hipMemcpy(b->d_wave + static_cast<size_t>(stream) * b->max_samples + b->n_samp[stream], wave,
                       static_cast<size_t>(n) * sizeof(float), hipMemcpyHostToDevice));
    b->n_samp[stream] += n;
  }
  if (input_finished) b->finished[stream] = 1;
  return KAMD_OK;
}

namespace kamd {
// segment i of the staged buffer -> its stream's waveform, appended (meta: dst offset, src offset, length per segment)
__global__ __launch_bounds__(256) void ScatterWaveKernel(const float *stage, float *wave, const int64_t *meta) {
  const int64_t dst = meta[3 * blockIdx.y], src = meta[3 * blockIdx.y + 1], len = meta[3 * blockIdx.y + 2];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < len; i += static_cast<int64_t>(gridDim.x) * 256) wave[dst + i] = stage[src + i];
}
}  // namespace kamd

// AcceptWaveform for many streams at once (a server's tick): ONE host-to-device copy of the concatenated chunks
// (waves[offsets[i] .. offsets[i+1]) belongs to streams[i]) and one scatter launch, instead of a copy per stream.
int kamd_stream_batch_accept_many(kamd_stream_batch *h, const int32_t *streams, int n, const float *waves, const int64_t *offsets,
                                  const int32_t *input_finished) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (n <= 0) return KAMD_OK;
  std::vector<int64_t> meta(3 * static_cast<size_t>(n));
  std::vector<char> seen(b->S, 0);
  int64_t longest = 0;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    const int64_t len = offsets[i + 1] - offsets[i];
    if (s < 0 || s >= b->S || !b->live[s]) return kamd::SetError(KAMD_ERR_ARG, "stream %d is not started", s);
    if (seen[s]) return kamd::SetError(KAMD_ERR_ARG, "stream %d listed twice", s);
    seen[s] = 1;
    if (len < 0) return kamd::SetError(KAMD_ERR_ARG, "stream %d: negative chunk length", s);
    if (b->finished[s] && len > 0) return kamd::SetError(KAMD_ERR_STATE, "AcceptWaveform called after InputFinished");
    if (b->n_samp[s] + len > b->max_samples) return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: more audio than max_seconds", s);
    meta[3 * i] = static_cast<int64_t>(s) * b->max_samples + b->n_samp[s];
    meta[3 * i + 1] = offsets[i] - offsets[0]; meta[3 * i + 2] = len;
    longest = std::max(longest, len);
  }
  const int64_t total = offsets[n] - offsets[0];
  if (total > 0) {
    const size_t need = static_cast<size_t>(total) * sizeof(float) + meta.size() * sizeof(int64_t) + 16;
    if (need > b->stage_cap) {
      if (b->d_stage) (void)hipFree(b->d_stage);
      b->d_stage = NULL; b->stage_cap = 0;
      KAMD_HIP(hipMalloc(&b->d_stage, 2 * need));
      b->stage_cap = 2 * need;
    }
    int64_t *d_meta = static_cast<int64_t *>(b->d_stage);
    float *d_st = reinterpret_cast<float *>(d_meta + meta.size() + 2);
    KAMD_HIP(hipMemcpy(d_meta, meta.data(), meta.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    KAMD_HIP(hipMemcpy(d_st, waves + offsets[0], static_cast<size_t>(total) * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kamd::ScatterWaveKernel, dim3(static_cast<unsigned>(std::min<int64_t>((longest + 255) / 256, 64)), n), dim3(256), 0, NULL,
                       d_st, b->d_wave, d_meta);
    KAMD_HIP(hipGetLastError());
    KAMD_HIP(hipStreamSynchronize(NULL));        // the staging buffer is reused by the next call
  }
  for (int i = 0; i < n; i++) {
    b->n_samp[streams[i]] += offsets[i + 1] - offsets[i];
    if (input_finished && input_finished[i]) b->finished[streams[i]] = 1;
  }
  return KAMD_OK;
}

// One tick: AdvanceDecoding for every listed stream.  frames_decoded[i] (optional) receives
// NumFramesDecoded of streams[i] after the tick.
int kamd_stream_batch_advance(kamd_stream_batch *h, const int32_t *streams, int n, int32_t *frames_decoded) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (n <= 0) return KAMD_OK;
  hipStream_t st = NULL;
  // ---- features: the frames that became computable (OnlineGenericBaseFeature::ComputeFeatures)
  std::vector<int64_t> wstart, wlen, frow;
  std::vector<int32_t> f0, fn;
  // validate everything before anything changes: a bad entry must not leave earlier streams half advanced
  {
    std::vector<char> seen(b->S, 0);
    for (int i = 0; i < n; i++) {
      const int s = streams[i];
      if (s < 0 || s >= b->S || !b->live[s]) return kamd::SetError(KAMD_ERR_ARG, "stream %d is not started", s);
      if (seen[s]) return kamd::SetError(KAMD_ERR_ARG, "stream %d listed twice in one tick", s);
      seen[s] = 1;
      if (b->status[s]) return kamd::SetError(KAMD_ERR_STATE, "stream %d failed earlier (flags %d): restart it (kamd_stream_batch_start)", s, b->status[s]);
      if (kamd_feat_num_frames_flush(b->feat, b->n_samp[s], b->finished[s] ? 1 : 0) > b->max_frames)
        return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: more frames than reserved", s);
    }
  }
  if (b->compact_at > 0.0f || b->prune_interval > 0) {
    // the usage figures are those of the sync that ended the streams' previous tick
    if (b->compacted_at.size() != static_cast<size_t>(b->S)) b->compacted_at.assign(b->S, 0);
    std::vector<int32_t> full;
    for (int i = 0; i < n; i++) {
      const int s = streams[i];
      int32_t tu = 0, tc = 1, lu = 0, lc = 1;
      if (kamd_decoder_lane_usage(b->dec, s, &tu, &tc, &lu, &lc) != KAMD_OK) return KAMD_ERR_ARG;
      if (b->compacted_at[s] > b->decoded[s]) b->compacted_at[s] = 0;          // (the stream was restarted)
      const bool arena = b->compact_at > 0.0f && (tu > b->compact_at * tc || lu > b->compact_at * lc);
      const bool due = b->prune_interval > 0 && b->decoded[s] - b->compacted_at[s] >= b->prune_interval;
      if (b->decoded[s] > 0 && (arena || due)) { full.push_back(s); b->compacted_at[s] = b->decoded[s]; }
    }
    if (!full.empty()) {
      int rc = kamd_decoder_compact(b->dec, full.data(), static_cast<int>(full.size()), st);
      if (rc != KAMD_OK) return rc;
      std::vector<int32_t> err(full.size(), 0);
      rc = kamd_decoder_sync_lanes(b->dec, full.data(), static_cast<int>(full.size()), err.data());
      if (rc != KAMD_OK) return rc;
      for (size_t k = 0; k < full.size(); k++)
        if (err[k]) {
          b->status[full[k]] = err[k];
          return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: compaction failed (flags %d)", full[k], err[k]);
        }
      b->n_compactions += static_cast<int64_t>(full.size());
    }
  }
  std::vector<int> new_frames(n);
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    const int ready = kamd_feat_num_frames_flush(b->feat, b->n_samp[s], b->finished[s] ? 1 : 0);
    new_frames[i] = std::max(ready, b->n_frames[s]);
    if (ready > b->n_frames[s]) {
      wstart.push_back(static_cast<int64_t>(s) * b->max_samples); wlen.push_back(b->n_samp[s]);
      f0.push_back(b->n_frames[s]); fn.push_back(ready - b->n_frames[s]);
      frow.push_back(static_cast<int64_t>(s) * b->max_frames + b->n_frames[s]);
    }
  }
  if (!wstart.empty()) {
    int rc = kamd_feat_compute_ranges_device(b->feat, b->d_wave, wstart.data(), wlen.data(), f0.data(), fn.data(),
                                             static_cast<int>(wstart.size()), b->d_frames, frow.data(), b->ld, st);
    if (rc != KAMD_OK) return rc;
  }
  for (int i = 0; i < n; i++) b->n_frames[streams[i]] = new_frames[i];      // the rows exist now
  // ---- online i-vectors: which chunks became computable, one estimate update per stream, new slots
  const int sub = kamd_nnet_frame_subsampling_factor(b->nnet);
  const int L = kamd_nnet_left_context(b->nnet), R = kamd_nnet_right_context(b->nnet);
  std::vector<int> ready_out(n, 0);
  // Between the incremental traceback (which advances the device's record of what the host knows) and the weighted
  // statistics update (which consumes the popped deltas) a failure leaves the two out of step for good: the tick's streams
  // are then out until restarted, like after a capacity error, instead of silently carrying wrong statistics.
  bool weighting_in_flight = false;
  auto lost = [&](int rc) {
    if (weighting_in_flight)
      for (int i = 0; i < n; i++) b->status[streams[i]] |= KAMD_STREAM_WEIGHTING_LOST;
    return rc;
  };
  if (b->ie && !b->sw.empty()) {
    // silence_weighting.ComputeCurrentTraceback(decoder.Decoder()); GetDeltaWeights(feature_pipeline.NumFramesReady());
    // IvectorFeature()->UpdateFrameWeights(delta_weights) -- one traceback launch for the tick's streams
    std::vector<int32_t> tl;
    int longest = 0;
    for (int i = 0; i < n; i++)
      if (b->decoded[streams[i]] > 0) { tl.push_back(streams[i]); longest = std::max(longest, b->decoded[streams[i]]); }
    std::vector<int32_t> tids(tl.size() * static_cast<size_t>(longest)), toks(tids.size()), cnt(tl.size()), n_new(tl.size());
    if (!tl.empty()) {
      int rc = kamd_decoder_frame_tracebacks_incremental(b->dec, tl.data(), static_cast<int>(tl.size()), tids.data(), toks.data(), longest,
                                                         cnt.data(), n_new.data());
      if (rc != KAMD_OK) return rc;
      weighting_in_flight = true;      // the device's "known" marks have advanced: from here a failure loses deltas
    }
    size_t k = 0;
    for (int i = 0; i < n; i++) {
      const int s = streams[i], F = b->n_frames[s];
      if (b->decoded[s] > 0) {
        if (cnt[k] != b->decoded[s])
          return lost(kamd::SetError(KAMD_ERR_STATE, "stream %d: best path covers %d of %d decoded frames", s, cnt[k], b->decoded[s]));
        int rc = kamd_silence_weighting_compute_traceback(b->sw[s], b->decoded[s], tids.data() + k * longest, toks.data() + k * longest, n_new[k]);
        if (rc != KAMD_OK) return lost(rc);
        k++;
      }
      const int iv_ready = b->finished[s] ? F : std::max(0, F - b->splice_right);
      int rc = kamd_silence_weighting_get_delta_weights(b->sw[s], iv_ready, NULL);
      if (rc != KAMD_OK) return lost(rc);
    }
  }
  if (b->ie) {
    std::vector<int64_t> u_row; std::vector<int32_t> u_base, u_done, u_upto, u_rec;
    std::vector<int> a_src, a_dst, a_cnt;
    struct Pending { int s, chunks, iv, slots; };
    std::vector<Pending> pend;                              // state changes, committed once the launches are issued
    for (int i = 0; i < n; i++) {
      const int s = streams[i], F = b->n_frames[s], fin = b->finished[s] ? 1 : 0, C = b->chunk;
      const int n_out_total = kamd_nnet_num_output_frames(b->nnet, F);
      int k = b->chunks_done[s];
      while (F > 0 && (fin ? k * C < n_out_total * sub : (k + 1) * C + R <= F)) k++;
      ready_out[i] = fin ? n_out_total : std::min(n_out_total, k * C / sub);
      if (k == b->chunks_done[s]) continue;
      // the estimate the decodable fetches: GetFrame(min(most recent input frame, NumFramesReady() - 1)) (decodable-online-looped.cc:174-183)
      const int iv_ready = fin ? F : std::max(0, F - b->splice_right);
      int src = b->S;                                     // "leave the iVector zero": row S of d_est
      Pending pd = {s, k, b->iv_done[s], b->slots_assigned[s]};
      if (iv_ready > b->iv_done[s]) {
        src = static_cast<int>(u_row.size());
        u_row.push_back(static_cast<int64_t>(s) * b->max_frames); u_base.push_back(F); u_done.push_back(b->iv_done[s]);
        u_upto.push_back(iv_ready); u_rec.push_back(s);
        pd.iv = iv_ready;
      } else if (b->iv_done[s] > 0) src = -1 - s;         // no new frame: the stream's last estimate (kept in its last slot)
      // slots the new chunks' input ranges introduce (nnet-compile-looped.cc:186-207)
      const int hi_t = k * C + R - 1;                     // last input time of chunk k - 1
      const int last_slot = (hi_t >= 0 ? hi_t / C : -1) - b->slot_first;
      const int have = b->slots_assigned[s];
      if (last_slot + 1 > b->max_slots) return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: i-vector slots exhausted", s);
      if (last_slot + 1 > have) {
        a_src.push_back(src); a_dst.push_back(s * b->max_slots + have); a_cnt.push_back(last_slot + 1 - have);
        pd.slots = last_slot + 1;
      }
      pend.push_back(pd);
    }
    // the extractor may be shared (another stream batch, an offline extraction between ticks): this batch's rows live in
    // its own workspace, bound for the duration of the call
    struct Bind {
      kamd_ivector_extractor *e;
      Bind(kamd_ivector_extractor *e_, kamd_ivector_workspace *w) : e(e_) { if (e) (void)kamd_ivector_extractor_bind_workspace(e, w); }
      ~Bind() { if (e) (void)kamd_ivector_extractor_bind_workspace(e, NULL); }
    };
    if (!u_row.empty() && !b->ie_ws) b->ie_ws = kamd_ivector_workspace_create();
    Bind bind(b->ie, u_row.empty() ? NULL : b->ie_ws);
    if (!u_row.empty() && b->sw.empty()) {
      int rc = kamd_ivector_stream_update_device(b->ie, b->d_frames, b->ld, static_cast<int64_t>(b->S) * b->max_frames, u_row.data(),
                                                 u_base.data(), u_done.data(), u_upto.data(), u_rec.data(), static_cast<int>(u_row.size()),
                                                 b->d_rec, b->d_est, st);
      if (rc != KAMD_OK) return rc;
    } else if (!u_row.empty()) {
      // UpdateStatsUntilFrameWeighted(upto - 1): the queued deltas up to that frame enter the statistics
      std::vector<int32_t> wl_off(1, 0), wl_frame; std::vector<float> wl_weight;
      for (size_t j = 0; j < u_row.size(); j++) {
        const size_t at = wl_frame.size();
        wl_frame.resize(at + u_upto[j]); wl_weight.resize(at + u_upto[j]);
        int32_t got = 0;
        int rc = kamd_silence_weighting_pop_until(b->sw[u_rec[j]], u_upto[j] - 1, wl_frame.data() + at, wl_weight.data() + at, u_upto[j], &got);
        if (rc != KAMD_OK) return lost(rc);
        weighting_in_flight = true;    // deltas have left the queue
        wl_frame.resize(at + got); wl_weight.resize(at + got);
        wl_off.push_back(static_cast<int32_t>(wl_frame.size()));
      }
      int rc = kamd_ivector_stream_update_weighted_device(b->ie, b->d_frames, b->ld, static_cast<int64_t>(b->S) * b->max_frames, u_row.data(),
                                                          u_base.data(), u_done.data(), u_upto.data(), u_rec.data(), wl_off.data(),
                                                          wl_frame.data(), wl_weight.data(), static_cast<int>(u_row.size()), b->d_rec, b->d_est, st);
      if (rc != KAMD_OK) return lost(rc);
      weighting_in_flight = false;     // issued: host and device agree again
    }
    if (!a_src.empty()) {
      const int m = static_cast<int>(a_src.size());
      // "no new frame" sources: copy from the stream's previous last slot (it holds the last estimate)
      std::vector<int> src_row(m);
      for (int j = 0; j < m; j++) src_row[j] = a_src[j];
      const float *src_base = b->d_est;
      // estimates and old slots live in different buffers: run two launches, one per source buffer
      std::vector<int> e_src, e_dst, e_cnt, o_src, o_dst, o_cnt;
      for (int j = 0; j < m; j++) {
        if (a_src[j] >= 0) { e_src.push_back(a_src[j]); e_dst.push_back(a_dst[j]); e_cnt.push_back(a_cnt[j]); }
        else { o_src.push_back(a_dst[j] - 1); o_dst.push_back(a_dst[j]); o_cnt.push_back(a_cnt[j]); }
      }
      for (int pass = 0; pass < 2; pass++) {
        std::vector<int> &S1 = pass ? o_src : e_src, &D1 = pass ? o_dst : e_dst, &C1 = pass ? o_cnt : e_cnt;
        const int mm = static_cast<int>(S1.size());
        if (mm == 0) continue;
        KAMD_HIP(hipMemcpyAsync(b->d_assign, S1.data(), mm * sizeof(int), hipMemcpyHostToDevice, st));
        KAMD_HIP(hipMemcpyAsync(b->d_assign + b->S, D1.data(), mm * sizeof(int), hipMemcpyHostToDevice, st));
        KAMD_HIP(hipMemcpyAsync(b->d_assign + 2 * b->S, C1.data(), mm * sizeof(int), hipMemcpyHostToDevice, st));
        KAMD_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(kamd::AssignSlotsKernel, dim3(mm), dim3(128), 0, st, pass ? b->d_slots : src_base, b->d_slots, b->d_assign,
                           b->d_assign + b->S, b->d_assign + 2 * b->S, b->iv_dim);
        KAMD_HIP(hipGetLastError());
        KAMD_HIP(hipStreamSynchronize(st));           // d_assign is reused by the second pass
      }
    }
    for (const Pending &pd : pend) { b->chunks_done[pd.s] = pd.chunks; b->iv_done[pd.s] = pd.iv; b->slots_assigned[pd.s] = pd.slots; }
  }
  // ---- nnet: one item per stream with new output frames (DecodableAmNnetLoopedOnline's rows)
  std::vector<int64_t> in_start, out_off;
  std::vector<int32_t> in_len;
  std::vector<kamd_decode_task> tasks;
  std::vector<int> k0s, counts, sid;
  std::vector<int32_t> sl_base, sl_first, sl_count, sl_t0;
  int64_t rows = 0;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    const int ready = b->ie ? ready_out[i] : kamd_nnet_num_frames_ready(b->nnet, b->n_frames[s], b->finished[s] ? 1 : 0);
    const int count = ready - b->decoded[s];
    if (count <= 0) continue;
    const int out_first = b->decoded[s];
    const int k0 = std::min(out_first, (L + sub - 1) / sub);          // leading outputs that only feed context
    const int in_first = sub * (out_first - k0);
    const int in_last = std::min(b->n_frames[s] - 1, sub * (out_first + count - 1) + R);
    const int T = in_last - in_first + 1;
    in_start.push_back(static_cast<int64_t>(s) * b->max_frames + in_first); in_len.push_back(T);
    out_off.push_back(rows);
    rows += kamd_nnet_num_output_frames(b->nnet, T);
    k0s.push_back(k0); counts.push_back(count); sid.push_back(s);
    if (b->ie) { sl_base.push_back(s * b->max_slots); sl_first.push_back(b->slot_first); sl_count.push_back(b->slots_assigned[s]); sl_t0.push_back(in_first); }
  }
  if (!sid.empty()) {
    const size_t need = static_cast<size_t>(rows) * b->P;
    if (need > b->ll_cap) {
      KAMD_HIP(hipDeviceSynchronize());
      if (b->d_ll) KAMD_HIP(hipFree(b->d_ll));
      b->d_ll = NULL; b->ll_cap = 0;
      KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&b->d_ll), 2 * need * sizeof(float)));
      b->ll_cap = 2 * need;
    }
    int rc = b->ie
      ? kamd_nnet_forward_slices_slots_device(b->nnet, b->d_frames, in_start.data(), in_len.data(), b->ld, b->d_slots, b->S * b->max_slots,
                                              b->chunk, sl_base.data(), sl_first.data(), sl_count.data(), sl_t0.data(),
                                              static_cast<int>(sid.size()), b->d_ll, out_off.data(), b->P, st)
      : kamd_nnet_forward_slices_device(b->nnet, b->d_frames, in_start.data(), in_len.data(), b->ld, NULL,
                                        static_cast<int>(sid.size()), b->d_ll, out_off.data(), b->P, st);
    if (rc != KAMD_OK) return rc;
    tasks.resize(sid.size());
    for (size_t k = 0; k < sid.size(); k++) {
      tasks[k].lane = sid[k]; tasks[k].n_frames = counts[k];
      tasks[k].d_loglikes = b->d_ll + static_cast<size_t>(out_off[k] + k0s[k]) * b->P;
      tasks[k].ld = b->P; tasks[k].reserved = 0;
    }
    rc = kamd_decoder_advance(b->dec, tasks.data(), static_cast<int>(tasks.size()), st);
    if (rc != KAMD_OK) return rc;
    // the lanes have consumed their rows whatever happens next: account for them first
    for (size_t k = 0; k < sid.size(); k++) b->decoded[sid[k]] += counts[k];
    std::vector<int32_t> lane_ids(sid.begin(), sid.end()), lane_err(sid.size(), 0);
    rc = kamd_decoder_sync_lanes(b->dec, lane_ids.data(), static_cast<int>(lane_ids.size()), lane_err.data());
    if (rc != KAMD_OK) return rc;
    int n_bad = 0, first_bad = -1;
    for (size_t k = 0; k < sid.size(); k++)
      if (lane_err[k]) { b->status[sid[k]] = lane_err[k]; n_bad++; if (first_bad < 0) first_bad = sid[k]; }
    if (frames_decoded) for (int i = 0; i < n; i++) frames_decoded[i] = b->decoded[streams[i]];
    if (n_bad)   // the other streams of the tick are fine and stay usable; kamd_stream_batch_get_status names the failed ones
      return kamd::SetError(KAMD_ERR_CAPACITY, "%d stream(s) of this tick exceeded their lane's capacity (first: stream %d, flags %d); "
                            "the others advanced normally", n_bad, first_bad, b->status[first_bad]);
    return KAMD_OK;
  }
  if (frames_decoded) for (int i = 0; i < n; i++) frames_decoded[i] = b->decoded[streams[i]];
  return KAMD_OK;
}

int kamd_stream_batch_get_status(const kamd_stream_batch *h, const int32_t *streams, int n, int32_t *status) {
  const StreamBatch *b = reinterpret_cast<const StreamBatch *>(h);
  for (int i = 0; i < n; i++) {
    if (streams[i] < 0 || streams[i] >= b->S) return kamd::SetError(KAMD_ERR_ARG, "stream %d out of range", streams[i]);
    status[i] = b->status[streams[i]];
  }
  return KAMD_OK;
}

int kamd_stream_batch_num_frames_ready(const kamd_stream_batch *h, int stream) {
  const StreamBatch *b = reinterpret_cast<const StreamBatch *>(h);
  if (stream < 0 || stream >= b->S) return 0;
  return b->n_frames[stream];
}

}  // extern "C"
