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
  float *d_tmp = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_tmp), static_cast<size_t>(n_out) * P * sizeof(float)));
  int64_t in_off[2] = {0, T}, out_off[1] = {0};
  int rc = kamd_nnet_forward_batch_device(n, d_feats + static_cast<size_t>(in_first) * ld_in, in_off, ld_in, NULL, 1,
                                          d_tmp, out_off, P, NULL);
  if (rc == KAMD_OK) {
    hipError_t e = hipMemcpy2D(d_out, ld_out * sizeof(float), d_tmp + static_cast<size_t>(k0) * P, P * sizeof(float),
                               P * sizeof(float), out_count, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "copy failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_tmp);
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
namespace kamd {
struct StreamBatch {
  kamd_feat *feat; kamd_nnet *nnet; kamd_decoder *dec;
  int S = 0, dim = 0, ld = 0, P = 0, max_frames = 0;
  int64_t max_samples = 0;
  float *d_wave = NULL, *d_frames = NULL, *d_ll = NULL;
  size_t ll_cap = 0;
  std::vector<int64_t> n_samp;
  std::vector<int> n_frames, decoded;
  std::vector<char> finished, live;
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
  b->finished.assign(max_streams, 0); b->live.assign(max_streams, 0);
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
  delete b;
}

// a new utterance on these streams: SingleUtteranceNnet3DecoderTpl's constructor calls
// decoder_.InitDecoding() (online-nnet3-decoding.cc:40)
int kamd_stream_batch_start(kamd_stream_batch *h, const int32_t *streams, int n) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    if (s < 0 || s >= b->S) return kamd::SetError(KAMD_ERR_ARG, "stream %d out of range", s);
    b->n_samp[s] = 0; b->n_frames[s] = 0; b->decoded[s] = 0; b->finished[s] = 0; b->live[s] = 1;
  }
  int rc = kamd_decoder_init(b->dec, streams, n, NULL);
  if (rc != KAMD_OK) return rc;
  return kamd_decoder_sync(b->dec);
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

// One tick: AdvanceDecoding for every listed stream.  frames_decoded[i] (optional) receives
// NumFramesDecoded of streams[i] after the tick.
int kamd_stream_batch_advance(kamd_stream_batch *h, const int32_t *streams, int n, int32_t *frames_decoded) {
  StreamBatch *b = reinterpret_cast<StreamBatch *>(h);
  if (n <= 0) return KAMD_OK;
  hipStream_t st = NULL;
  // ---- features: the frames that became computable (OnlineGenericBaseFeature::ComputeFeatures)
  std::vector<int64_t> wstart, wlen, frow;
  std::vector<int32_t> f0, fn;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    if (s < 0 || s >= b->S || !b->live[s]) return kamd::SetError(KAMD_ERR_ARG, "stream %d is not started", s);
    const int ready = kamd_feat_num_frames_flush(b->feat, b->n_samp[s], b->finished[s] ? 1 : 0);
    if (ready > b->max_frames) return kamd::SetError(KAMD_ERR_CAPACITY, "stream %d: more frames than reserved", s);
    if (ready > b->n_frames[s]) {
      wstart.push_back(static_cast<int64_t>(s) * b->max_samples); wlen.push_back(b->n_samp[s]);
      f0.push_back(b->n_frames[s]); fn.push_back(ready - b->n_frames[s]);
      frow.push_back(static_cast<int64_t>(s) * b->max_frames + b->n_frames[s]);
      b->n_frames[s] = ready;
    }
  }
  if (!wstart.empty()) {
    int rc = kamd_feat_compute_ranges_device(b->feat, b->d_wave, wstart.data(), wlen.data(), f0.data(), fn.data(),
                                             static_cast<int>(wstart.size()), b->d_frames, frow.data(), b->ld, st);
    if (rc != KAMD_OK) return rc;
  }
  // ---- nnet: one item per stream with new output frames (DecodableAmNnetLoopedOnline's rows)
  const int sub = kamd_nnet_frame_subsampling_factor(b->nnet);
  const int L = kamd_nnet_left_context(b->nnet), R = kamd_nnet_right_context(b->nnet);
  std::vector<int64_t> in_start, out_off;
  std::vector<int32_t> in_len;
  std::vector<kamd_decode_task> tasks;
  std::vector<int> k0s, counts, sid;
  int64_t rows = 0;
  for (int i = 0; i < n; i++) {
    const int s = streams[i];
    const int ready = kamd_nnet_num_frames_ready(b->nnet, b->n_frames[s], b->finished[s] ? 1 : 0);
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
    int rc = kamd_nnet_forward_slices_device(b->nnet, b->d_frames, in_start.data(), in_len.data(), b->ld, NULL,
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
    rc = kamd_decoder_sync(b->dec);
    if (rc != KAMD_OK) return rc;
    for (size_t k = 0; k < sid.size(); k++) b->decoded[sid[k]] += counts[k];
  }
  if (frames_decoded) for (int i = 0; i < n; i++) frames_decoded[i] = b->decoded[streams[i]];
  return KAMD_OK;
}

int kamd_stream_batch_num_frames_ready(const kamd_stream_batch *h, int stream) {
  const StreamBatch *b = reinterpret_cast<const StreamBatch *>(h);
  if (stream < 0 || stream >= b->S) return 0;
  return b->n_frames[stream];
}

}  // extern "C"
