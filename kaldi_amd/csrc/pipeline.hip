// pipeline.hip -- wav -> MFCC -> TDNN-F log-likelihoods -> lattice, device resident.
//
// Mirrors the per-utterance flow of nnet3-latgen-faster-batch
// (nnet3bin/nnet3-latgen-faster-batch.cc:170-214: AcceptInput -> NnetBatchComputer ->
// LatticeFasterDecoder::AdvanceDecoding -> GetRawLattice) for a whole batch of
// utterances at once.  Nothing crosses PCIe between the stages: features are written
// straight into the (zero padded) GEMM operand, log-likelihoods stay in HBM and the
// decoder lanes read them in place (the reference copies per chunk:
// nnet-am-decodable-simple.cc:256,274; nnet-batch-compute.cc:412,469).
#include <vector>

#include "common.h"

namespace kamd {
struct Pipeline {
  kamd_feat *feat; kamd_nnet *nnet; kamd_decoder *dec;
  int n_utts = 0, feat_dim = 0, ld_feat = 0, P = 0;
  std::vector<int64_t> wave_off, feat_off, out_off;
  float *d_waves = NULL, *d_feats = NULL, *d_ll = NULL, *d_iv = NULL;
  size_t waves_cap = 0, feats_cap = 0, ll_cap = 0, iv_cap = 0;
  int iv_dim = 0;
  bool have_feats = false;      // features were handed in (kamd_pipeline_load_features): skip the feature stage
  // online ivectors (one row per ivector_period frames) instead of one ivector per utterance
  float *d_oiv = NULL; size_t oiv_cap = 0; std::vector<int64_t> oiv_off; int oiv_period = 0, frames_per_chunk = 50;
  hipEvent_t ev[5];
};
template <typename T>
static int GrowBuf(T **p, size_t *cap, size_t need) {
  if (need <= *cap) return KAMD_OK;
  if (*p) (void)hipFree(*p);
  *p = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(p), need * sizeof(T)));
  *cap = need;
  return KAMD_OK;
}
}  // namespace kamd
using kamd::Pipeline;

extern "C" {

kamd_pipeline *kamd_pipeline_create(kamd_feat *feat, kamd_nnet *nnet, kamd_decoder *dec) {
  Pipeline *p = new Pipeline();
  p->feat = feat; p->nnet = nnet; p->dec = dec;
  p->feat_dim = kamd_feat_dim(feat);
  p->ld_feat = kamd::RoundUp(p->feat_dim, 16);
  p->P = kamd_nnet_output_dim(nnet);
  for (int i = 0; i < 5; i++)
    if (hipEventCreate(&p->ev[i]) != hipSuccess) { kamd::SetError(KAMD_ERR_HIP, "hipEventCreate failed"); delete p; return NULL; }
  return reinterpret_cast<kamd_pipeline *>(p);
}

void kamd_pipeline_destroy(kamd_pipeline *h) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (!p) return;
  if (p->d_waves) (void)hipFree(p->d_waves);
  if (p->d_feats) (void)hipFree(p->d_feats);
  if (p->d_ll) (void)hipFree(p->d_ll);
  if (p->d_iv) (void)hipFree(p->d_iv);
  if (p->d_oiv) (void)hipFree(p->d_oiv);
  for (int i = 0; i < 5; i++) (void)hipEventDestroy(p->ev[i]);
  delete p;
}

int kamd_pipeline_load_batch(kamd_pipeline *h, const float *waves, const int64_t *h_wave_off, int n_utts) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (n_utts <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty batch");
  p->n_utts = n_utts; p->have_feats = false;
  p->wave_off.assign(h_wave_off, h_wave_off + n_utts + 1);
  p->feat_off.assign(n_utts + 1, 0); p->out_off.assign(n_utts + 1, 0);
  for (int u = 0; u < n_utts; u++) {
    int T = kamd_feat_num_frames(p->feat, h_wave_off[u + 1] - h_wave_off[u]);
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d too short for one frame", u);
    p->feat_off[u + 1] = p->feat_off[u] + T;
    p->out_off[u + 1] = p->out_off[u] + kamd_nnet_num_output_frames(p->nnet, T);
  }
  const size_t ns = static_cast<size_t>(h_wave_off[n_utts] - h_wave_off[0]);
  if (kamd::GrowBuf(&p->d_waves, &p->waves_cap, ns) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowBuf(&p->d_feats, &p->feats_cap, static_cast<size_t>(p->feat_off[n_utts]) * p->ld_feat) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowBuf(&p->d_ll, &p->ll_cap, static_cast<size_t>(p->out_off[n_utts]) * p->P) != KAMD_OK) return KAMD_ERR_HIP;
  KAMD_HIP(hipMemcpy(p->d_waves, waves + h_wave_off[0], ns * sizeof(float), hipMemcpyHostToDevice));
  if (h_wave_off[0] != 0) for (int u = 0; u <= n_utts; u++) p->wave_off[u] -= h_wave_off[0];
  return KAMD_OK;
}

// The features-rspecifier case of nnet3-latgen-faster (nnet3bin/nnet3-latgen-faster.cc:166-200: a
// SequentialBaseFloatMatrixReader of features some other tool computed): utterance u owns rows
// [h_row_off[u], h_row_off[u+1]) of `feats` (row-major, `dim` columns = the model's input dim).
int kamd_pipeline_load_features(kamd_pipeline *h, const float *feats, const int64_t *h_row_off, int n_utts, int dim) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (n_utts <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty batch");
  if (dim != p->feat_dim) return kamd::SetError(KAMD_ERR_ARG, "feature dim %d, the pipeline's feature stage has %d", dim, p->feat_dim);
  p->n_utts = n_utts; p->have_feats = true;
  p->feat_off.assign(n_utts + 1, 0); p->out_off.assign(n_utts + 1, 0);
  for (int u = 0; u < n_utts; u++) {
    const int64_t T = h_row_off[u + 1] - h_row_off[u];
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames", u);
    p->feat_off[u + 1] = p->feat_off[u] + T;
    p->out_off[u + 1] = p->out_off[u] + kamd_nnet_num_output_frames(p->nnet, static_cast<int>(T));
  }
  const size_t rows = static_cast<size_t>(p->feat_off[n_utts]);
  if (kamd::GrowBuf(&p->d_feats, &p->feats_cap, rows * p->ld_feat) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowBuf(&p->d_ll, &p->ll_cap, static_cast<size_t>(p->out_off[n_utts]) * p->P) != KAMD_OK) return KAMD_ERR_HIP;
  if (p->ld_feat != dim) KAMD_HIP(hipMemset(p->d_feats, 0, rows * p->ld_feat * sizeof(float)));
  KAMD_HIP(hipMemcpy2D(p->d_feats, p->ld_feat * sizeof(float), feats + h_row_off[0] * dim, dim * sizeof(float),
                       dim * sizeof(float), rows, hipMemcpyHostToDevice));
  return KAMD_OK;
}

int kamd_pipeline_set_ivectors(kamd_pipeline *h, const float *ivectors, int dim) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (dim <= 0 || !ivectors) { p->iv_dim = 0; return KAMD_OK; }
  if (p->n_utts <= 0) return kamd::SetError(KAMD_ERR_STATE, "load the batch before its ivectors");
  const size_t n = static_cast<size_t>(p->n_utts) * dim;
  if (kamd::GrowBuf(&p->d_iv, &p->iv_cap, n) != KAMD_OK) return KAMD_ERR_HIP;
  KAMD_HIP(hipMemcpy(p->d_iv, ivectors, n * sizeof(float), hipMemcpyHostToDevice));
  p->iv_dim = dim; p->oiv_period = 0;
  return KAMD_OK;
}

// --online-ivectors / --online-ivector-period of nnet3-latgen-faster: utterance u (in the order of
// the loaded batch) owns rows [h_row_off[u], h_row_off[u+1]) of `ivectors`; the nnet stage then
// runs chunk by chunk like DecodableNnetSimple (kamd_nnet_forward_chunked_device).
int kamd_pipeline_set_online_ivectors(kamd_pipeline *h, const float *ivectors, const int64_t *h_row_off, int dim,
                                      int ivector_period, int frames_per_chunk) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (dim <= 0 || !ivectors) { p->oiv_period = 0; p->oiv_off.clear(); return KAMD_OK; }
  if (p->n_utts <= 0) return kamd::SetError(KAMD_ERR_STATE, "load the batch before its ivectors");
  if (ivector_period <= 0 || frames_per_chunk <= 0) return kamd::SetError(KAMD_ERR_ARG, "bad ivector period / frames per chunk");
  const size_t n = static_cast<size_t>(h_row_off[p->n_utts] - h_row_off[0]) * dim;
  if (kamd::GrowBuf(&p->d_oiv, &p->oiv_cap, n) != KAMD_OK) return KAMD_ERR_HIP;
  KAMD_HIP(hipMemcpy(p->d_oiv, ivectors + h_row_off[0] * dim, n * sizeof(float), hipMemcpyHostToDevice));
  p->oiv_off.assign(h_row_off, h_row_off + p->n_utts + 1);
  for (int u = p->n_utts; u >= 0; u--) p->oiv_off[u] -= h_row_off[0];
  p->iv_dim = dim; p->oiv_period = ivector_period; p->frames_per_chunk = frames_per_chunk;
  return KAMD_OK;
}

int kamd_pipeline_run(kamd_pipeline *h, float stage_ms[4]) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  const int n = p->n_utts;
  if (n <= 0) return kamd::SetError(KAMD_ERR_STATE, "no batch loaded");
  hipStream_t st = NULL;
  KAMD_HIP(hipEventRecord(p->ev[0], st));
  int rc = KAMD_OK;
  if (!p->have_feats)
    rc = kamd_feat_compute_batch_device(p->feat, p->d_waves, p->wave_off.data(), n, p->d_feats,
                                        p->feat_off.data(), p->ld_feat, st);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[1], st));
  if (p->oiv_period > 0)
    rc = kamd_nnet_forward_chunked_device(p->nnet, p->d_feats, p->feat_off.data(), p->ld_feat, p->d_oiv, p->oiv_off.data(),
                                          p->iv_dim, p->oiv_period, p->frames_per_chunk, n, p->d_ll, p->out_off.data(), p->P, st);
  else
    rc = kamd_nnet_forward_batch_device(p->nnet, p->d_feats, p->feat_off.data(), p->ld_feat, p->iv_dim > 0 ? p->d_iv : NULL, n,
                                        p->d_ll, p->out_off.data(), p->P, st);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[2], st));
  std::vector<int32_t> lanes(n);
  std::vector<kamd_decode_task> tasks(n);
  for (int u = 0; u < n; u++) {
    lanes[u] = u;
    tasks[u].lane = u; tasks[u].n_frames = static_cast<int>(p->out_off[u + 1] - p->out_off[u]);
    tasks[u].d_loglikes = p->d_ll + static_cast<size_t>(p->out_off[u]) * p->P;
    tasks[u].ld = p->P; tasks[u].reserved = 0;
  }
  {
    std::vector<int32_t> fr(n);
    for (int u = 0; u < n; u++) fr[u] = tasks[u].n_frames;
    rc = kamd_decoder_reserve(p->dec, fr.data(), n);
    if (rc != KAMD_OK) return rc;
  }
  rc = kamd_decoder_init(p->dec, lanes.data(), n, st);
  if (rc != KAMD_OK) return rc;
  rc = kamd_decoder_advance(p->dec, tasks.data(), n, st);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[3], st));
  rc = kamd_decoder_finalize(p->dec, lanes.data(), n, st);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[4], st));
  rc = kamd_decoder_sync(p->dec);
  if (stage_ms) {
    for (int i = 0; i < 4; i++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
      stage_ms[i] = ms;
    }
  }
  return rc;
}

int kamd_pipeline_get_loglikes(kamd_pipeline *h, int utt, float *out, int rows_cap, int *rows, int *cols) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (utt < 0 || utt >= p->n_utts) return kamd::SetError(KAMD_ERR_ARG, "bad utterance index");
  const int r = static_cast<int>(p->out_off[utt + 1] - p->out_off[utt]);
  *rows = r; *cols = p->P;
  if (r > rows_cap) return kamd::SetError(KAMD_ERR_ARG, "buffer too small");
  KAMD_HIP(hipMemcpy(out, p->d_ll + static_cast<size_t>(p->out_off[utt]) * p->P,
                     static_cast<size_t>(r) * p->P * sizeof(float), hipMemcpyDeviceToHost));
  return KAMD_OK;
}

int kamd_pipeline_get_features(kamd_pipeline *h, int utt, float *out, int rows_cap, int *rows, int *cols) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (utt < 0 || utt >= p->n_utts) return kamd::SetError(KAMD_ERR_ARG, "bad utterance index");
  const int r = static_cast<int>(p->feat_off[utt + 1] - p->feat_off[utt]);
  *rows = r; *cols = p->feat_dim;
  if (r > rows_cap) return kamd::SetError(KAMD_ERR_ARG, "buffer too small");
  KAMD_HIP(hipMemcpy2D(out, p->feat_dim * sizeof(float),
                       p->d_feats + static_cast<size_t>(p->feat_off[utt]) * p->ld_feat, p->ld_feat * sizeof(float),
                       p->feat_dim * sizeof(float), r, hipMemcpyDeviceToHost));
  return KAMD_OK;
}

}  // extern "C"
