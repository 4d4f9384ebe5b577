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

}  // extern "C"
