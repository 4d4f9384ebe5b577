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
  kamd_ivector_extractor *iv_extractor = NULL;   // i-vectors estimated from the batch's own features
  bool have_feats = false;      // features were handed in (kamd_pipeline_load_features): skip the feature stage
  // online ivectors (one row per ivector_period frames) instead of one ivector per utterance
  float *d_oiv = NULL; size_t oiv_cap = 0; std::vector<int64_t> oiv_off; int oiv_period = 0, frames_per_chunk = 50;
  hipEvent_t ev[5];
  // nnet / search overlap (kamd_pipeline_set_overlap)
  std::vector<int32_t> bounds;
  hipStream_t s_nnet = NULL, s_dec = NULL;
  std::vector<hipEvent_t> slice_done;
  float *d_tmp = NULL; size_t tmp_cap = 0;
  int64_t *d_desc = NULL; size_t desc_cap = 0;
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
  for (hipEvent_t e : p->slice_done) (void)hipEventDestroy(e);
  if (p->s_nnet) (void)hipStreamDestroy(p->s_nnet);
  if (p->s_dec) (void)hipStreamDestroy(p->s_dec);
  if (p->d_tmp) (void)hipFree(p->d_tmp);
  if (p->d_desc) (void)hipFree(p->d_desc);
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
  if (p->ld_feat != dim) { KAMD_HIP(hipMemset(p->d_feats, 0, rows * p->ld_feat * sizeof(float))); KAMD_HIP(hipDeviceSynchronize()); }
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

int kamd_pipeline_set_ivector_extractor(kamd_pipeline *h, kamd_ivector_extractor *e, int frames_per_chunk) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (!e) { p->iv_extractor = NULL; p->oiv_period = 0; p->iv_dim = 0; p->oiv_off.clear(); return KAMD_OK; }
  if (frames_per_chunk <= 0) return kamd::SetError(KAMD_ERR_ARG, "bad frames per chunk");
  p->iv_extractor = e; p->frames_per_chunk = frames_per_chunk;
  return KAMD_OK;
}

int kamd_pipeline_set_overlap(kamd_pipeline *h, const int32_t *bounds, int n_bounds) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  if (n_bounds < 0 || n_bounds > 6) return kamd::SetError(KAMD_ERR_ARG, "0 .. 6 slice boundaries");
  for (int i = 0; i < n_bounds; i++)
    if (bounds[i] <= (i ? bounds[i - 1] : 0)) return kamd::SetError(KAMD_ERR_ARG, "slice boundaries must be positive and increasing");
  p->bounds.assign(bounds, bounds + n_bounds);
  if (n_bounds > 0 && !p->s_nnet) {
    // streams of different priorities: two streams of ONE priority can share a hardware queue, on which their kernels run
    // in submission order (seen in batch.cc's long-utterance decoder: kamd_batch_decoder_set_long_decoder)
    int least = 0, greatest = 0;
    KAMD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    KAMD_HIP(hipStreamCreateWithFlags(&p->s_nnet, hipStreamNonBlocking));
    KAMD_HIP(hipStreamCreateWithPriority(&p->s_dec, hipStreamNonBlocking, greatest));
  }
  while (static_cast<int>(p->slice_done.size()) < n_bounds + 1) {
    hipEvent_t e;
    KAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    p->slice_done.push_back(e);
  }
  return KAMD_OK;
}

// kamd_pipeline_run with the nnet stage cut in time (kamd_pipeline_set_overlap).  Stream s_nnet:
// features, then per slice one batched forward over (utterance, output-frame range) items -- each
// with the left / right context it needs, clamped only at the utterance's own edges, computed into
// a scratch matrix -- and a row copy of the frames the slice owns into the log-likelihood matrix.
// Stream s_dec: InitDecoding, then per slice "wait for the slice, AdvanceDecoding over its frames".
// stage_ms: [0] features, [1] the first slice's forward (all the search waits for), [2] first
// AdvanceKernel launch .. last one's end (the later forwards run inside it), [3] finalize.
static int RunOverlapped(Pipeline *p, float stage_ms[4]) {
  const int n = p->n_utts, sub = kamd_nnet_frame_subsampling_factor(p->nnet);
  const int Lc = kamd_nnet_left_context(p->nnet), Rc = kamd_nnet_right_context(p->nnet);
  const int kctx = (Lc + sub - 1) / sub;
  hipStream_t s1 = p->s_nnet, s2 = p->s_dec;
  std::vector<int32_t> lanes(n), fr(n);
  for (int u = 0; u < n; u++) { lanes[u] = u; fr[u] = static_cast<int>(p->out_off[u + 1] - p->out_off[u]); }
  int rc = kamd_decoder_reserve(p->dec, fr.data(), n);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[0], s1));
  if (!p->have_feats)
    rc = kamd_feat_compute_batch_device(p->feat, p->d_waves, p->wave_off.data(), n, p->d_feats, p->feat_off.data(), p->ld_feat, s1);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[1], s1));
  rc = kamd_decoder_init(p->dec, lanes.data(), n, s2);
  if (rc != KAMD_OK) return rc;
  const int n_slices = static_cast<int>(p->bounds.size()) + 1;
  // scratch: the largest slice's items (rows incl. context) x P
  for (int s = 0; s < n_slices; s++) {
    const int b0 = s ? p->bounds[s - 1] : 0, b1 = s < n_slices - 1 ? p->bounds[s] : 2147483647;
    std::vector<int64_t> in_start, tmp_off, src_row, dst_row;
    std::vector<int32_t> in_len, cnt;
    std::vector<kamd_decode_task> tasks;
    int64_t tmp_rows = 0;
    int max_cnt = 1;
    for (int u = 0; u < n; u++) {
      const int T = static_cast<int>(p->feat_off[u + 1] - p->feat_off[u]), n_out = fr[u];
      if (n_out <= b0) continue;
      const int o1 = std::min(n_out, b1), num = o1 - b0;
      const int k0 = std::min(b0, kctx);
      const int in_first = sub * (b0 - k0), in_last = std::min(T - 1, (o1 - 1) * sub + Rc);
      in_start.push_back(p->feat_off[u] + in_first); in_len.push_back(in_last - in_first + 1);
      tmp_off.push_back(tmp_rows);
      src_row.push_back(tmp_rows + k0); dst_row.push_back(p->out_off[u] + b0); cnt.push_back(num);
      tmp_rows += (in_last - in_first + 1 + sub - 1) / sub;
      max_cnt = std::max(max_cnt, num);
      kamd_decode_task t;
      t.lane = u; t.n_frames = num; t.d_loglikes = p->d_ll + static_cast<size_t>(p->out_off[u] + b0) * p->P; t.ld = p->P; t.reserved = 0;
      tasks.push_back(t);
    }
    const int n_items = static_cast<int>(in_start.size());
    if (n_items == 0) break;
    if (kamd::GrowBuf(&p->d_tmp, &p->tmp_cap, static_cast<size_t>(tmp_rows) * p->P) != KAMD_OK) return KAMD_ERR_HIP;
    if (kamd::GrowBuf(&p->d_desc, &p->desc_cap, static_cast<size_t>(n_slices) * (3 * n + 8)) != KAMD_OK) return KAMD_ERR_HIP;
    // a slice's descriptors live in their own part of d_desc: the copy kernel of slice s may
    // still be queued when the host writes those of slice s + 1
    int64_t *d_src = p->d_desc + static_cast<size_t>(s) * (3 * n + 8), *d_dst = d_src + n;
    int *d_cnt = reinterpret_cast<int *>(d_dst + n);
    // the scratch matrix is reused by the next slice's forward: stream order on s1 (forward s,
    // copy s, forward s + 1) keeps that safe
    rc = kamd_nnet_forward_slices_device(p->nnet, p->d_feats, in_start.data(), in_len.data(), p->ld_feat, NULL, n_items, p->d_tmp,
                                         tmp_off.data(), p->P, s1);
    if (rc != KAMD_OK) return rc;
    KAMD_HIP(hipMemcpyAsync(d_src, src_row.data(), n_items * 8, hipMemcpyHostToDevice, s1));
    KAMD_HIP(hipMemcpyAsync(d_dst, dst_row.data(), n_items * 8, hipMemcpyHostToDevice, s1));
    KAMD_HIP(hipMemcpyAsync(d_cnt, cnt.data(), n_items * 4, hipMemcpyHostToDevice, s1));
    rc = kamd::CopyRowBlocks(p->d_tmp, p->P, p->d_ll, p->P, d_src, d_dst, d_cnt, n_items, max_cnt, p->P, s1);
    if (rc != KAMD_OK) return rc;
    KAMD_HIP(hipEventRecord(p->slice_done[s], s1));
    if (s == 0) KAMD_HIP(hipEventRecord(p->ev[2], s1));
    KAMD_HIP(hipStreamSynchronize(s1));   // the descriptor vectors above are pageable host memory; the search needs the slice anyway
    KAMD_HIP(hipStreamWaitEvent(s2, p->slice_done[s], 0));
    rc = kamd_decoder_advance(p->dec, tasks.data(), static_cast<int>(tasks.size()), s2);
    if (rc != KAMD_OK) return rc;
  }
  KAMD_HIP(hipEventRecord(p->ev[3], s2));
  rc = kamd_decoder_finalize(p->dec, lanes.data(), n, s2);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(p->ev[4], s2));
  rc = kamd_decoder_sync(p->dec);
  KAMD_HIP(hipStreamSynchronize(s1));
  if (stage_ms) {
    for (int i = 0; i < 4; i++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
      stage_ms[i] = ms;
    }
  }
  return rc;
}

int kamd_pipeline_run(kamd_pipeline *h, float stage_ms[4]) {
  Pipeline *p = reinterpret_cast<Pipeline *>(h);
  const int n = p->n_utts;
  if (n <= 0) return kamd::SetError(KAMD_ERR_STATE, "no batch loaded");
  if (!p->bounds.empty() && p->iv_dim == 0 && p->oiv_period == 0 && !p->iv_extractor) return RunOverlapped(p, stage_ms);
  hipStream_t st = NULL;
  KAMD_HIP(hipEventRecord(p->ev[0], st));
  int rc = KAMD_OK;
  if (!p->have_feats)
    rc = kamd_feat_compute_batch_device(p->feat, p->d_waves, p->wave_off.data(), n, p->d_feats,
                                        p->feat_off.data(), p->ld_feat, st);
  if (rc != KAMD_OK) return rc;
  if (p->iv_extractor) {
    // OnlineIvectorFeature on the features just computed, straight into the matrix the chunked
    // forward reads its per-chunk i-vectors from (counted with the feature stage)
    p->oiv_off.assign(n + 1, 0);
    for (int u = 0; u < n; u++)
      p->oiv_off[u + 1] = p->oiv_off[u] + kamd_ivector_num_ivectors(p->iv_extractor, static_cast<int>(p->feat_off[u + 1] - p->feat_off[u]));
    p->iv_dim = kamd_ivector_dim(p->iv_extractor); p->oiv_period = kamd_ivector_period(p->iv_extractor);
    if (kamd::GrowBuf(&p->d_oiv, &p->oiv_cap, static_cast<size_t>(p->oiv_off[n]) * p->iv_dim) != KAMD_OK) return KAMD_ERR_HIP;
    rc = kamd_ivector_extract_online_device(p->iv_extractor, p->d_feats, p->feat_off.data(), p->ld_feat, n, p->d_oiv, p->oiv_off.data(), st);
    if (rc != KAMD_OK) return rc;
  }
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
