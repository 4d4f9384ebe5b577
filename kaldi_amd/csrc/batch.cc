// batch.cc -- a whole test set through the device: NnetBatchDecoder's job.
//
// Reference: nnet3/nnet-batch-compute.h:606-833 (NnetBatchDecoder: AcceptInput :665,
// Finished :690, GetOutput :700; .cc:1007-1343: one compute thread feeding N decoder
// threads, each of which decodes, determinizes and queues its output) and the binary built
// on it, nnet3bin/nnet3-latgen-faster-batch.cc:170-214; decode.sh's --nj split
// (steps/nnet3/decode.sh:96,123) is the same idea across processes.
//
// MI355X shape of the same pipeline, per GPU:
//   1. every waveform of the shard is resident in HBM (288 GB: LibriSpeech test-clean is
//      1.2 GB of float samples, its log-likelihood matrix 15 GB);
//   2. features for all utterances in one launch, the acoustic model in a few passes of
//      ~4e5 input frames (large M for the MFMA GEMMs, bounded activation memory);
//   3. ONE work-queue launch of the decoder (kamd_decoder_queue_*): a lane per compute unit,
//      utterances handed out longest first, finalize + lattice hand-off inside the kernel;
//   4. the host tail -- D2H of each finished lattice on a copy stream, canonical numbering,
//      GetBestPath, DeterminizeLatticePhonePrunedWrapper -- on a pool of host threads WHILE
//      the kernel is still decoding the rest (decoder-wrappers.cc:201-296 is the per-utterance
//      recipe).  Host code only; all device work goes through the C-ABI of the stages.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#include "common.h"

namespace kamd {

struct UttOut {
  int done = 0, status = KAMD_OK;     // status: KAMD_OK, or the error of this utterance alone
  std::string message;
  kamd_queue_result rec;
  bool have_path = false;
  std::vector<int32_t> words, ali;
  float graph_cost = INFINITY, ac_cost = INFINITY;
  // raw lattice (canonical), when kept
  int32_t num_states = 0, num_arcs = 0, start = -1;
  int32_t *st_frame = NULL, *st_hclg = NULL; float *st_cost = NULL, *st_final = NULL; kamd_lat_arc *arcs = NULL;
  kamd_compact_lattice *clat = NULL;
  double host_ms = 0;
  void Clear() {
    free(st_frame); free(st_hclg); free(st_cost); free(st_final); free(arcs);
    st_frame = st_hclg = NULL; st_cost = st_final = NULL; arcs = NULL;
    if (clat) kamd_compact_lattice_destroy(clat);
    clat = NULL; done = 0; status = KAMD_OK; have_path = false; words.clear(); ali.clear(); num_states = num_arcs = 0; start = -1;
    host_ms = 0; message.clear();
  }
};

struct BatchDecoder {
  kamd_feat *feat; kamd_nnet *nnet; kamd_decoder *dec;
  kamd_batch_opts opts;
  std::vector<int32_t> tid_phone;
  int device = 0;
  int feat_dim = 0, ld_feat = 0, P = 0;
  int n_utts = 0;
  // utterances too short for one frame are not decoded: they fail alone ("Zero-length utterance", nnet3-latgen-faster-
  // batch.cc:184-188).  kept[k] = the caller's index of the k-th utterance that is; every offset array below is indexed by k
  std::vector<int> kept, skipped;
  std::vector<int64_t> wave_off, feat_off, out_off;
  float *d_waves = NULL, *d_feats = NULL, *d_ll = NULL, *d_iv = NULL;
  size_t waves_cap = 0, feats_cap = 0, ll_cap = 0, iv_cap = 0;
  bool from_features = false, have_iv = false;   // the loaded set: feature matrices (+ i-vectors) instead of waveforms
  // online i-vectors estimated from the set's own features (steps/nnet3/decode.sh:105-107 --online-ivectors + --online-
  // ivector-period; the matrices extract_ivectors_online.sh would have written): kamd_batch_decoder_set_ivector_extractor.
  // The acoustic model then runs chunk by chunk like DecodableNnetSimple (kamd_nnet_forward_chunked_device).
  kamd_ivector_extractor *iv_extractor = NULL;
  int frames_per_chunk = 50;
  int chunk_rule = 0;                            // 0: DecodableNnetSimple's chunks, 1: NnetBatchComputer's tasks (kamd_batch_decoder_set_chunk_rule)
  float *d_oiv = NULL; size_t oiv_cap = 0;
  std::vector<int64_t> oiv_off;
  // kamd_batch_decoder_load_host: the samples stay in the caller's memory and every run() uploads them pass by pass
  const float *h_waves = NULL;                   // caller's buffer (kept utterance k starts at h_wave_src[k])
  bool host_unloaded = false;                    // kamd_batch_decoder_unload_host: the samples are gone, run() needs a new load
  void *h_registered = NULL;                     // ... page-locked in place by load_host (hipHostRegister) when the runtime allows it:
                                                 // the copies then read it directly; otherwise they go through h_stage
  uintptr_t reg_lo = 0, reg_hi = 0;              // the locked bytes: the pages that lie wholly inside the caller's samples
  std::vector<int64_t> h_wave_src;
  std::vector<int> pass_u0;                      // pass p = kept utterances [pass_u0[p], pass_u0[p + 1])
  std::vector<int64_t> pass_frames;              // feature frames of pass p
  int64_t *d_feat_meta = NULL; size_t feat_meta_cap = 0;
  std::vector<size_t> feat_meta_off;             // first word of pass p's offsets in d_feat_meta
  static const int kStageBufs = 4;
  static const size_t kStageFloats = 8u << 20;   // 32 MB each
  float *h_stage[kStageBufs] = {};
  hipEvent_t ev_stage[kStageBufs] = {};
  hipStream_t s_up = NULL;
  std::vector<hipEvent_t> ev_up, ev_f0, ev_f1, ev_n1;   // per pass: copies done; features start / end; model end
  std::vector<hipEvent_t> ev_iv0, ev_iv1, ev_n0;        // per pass: i-vector extraction start / end; chunked model start
  hipEvent_t ev_solve[2] = {};                          // the one solver launch of a run with the solver deferred
  int64_t iv_reserved = 0;                              // i-vector rows whose step statistics the extractor holds buffers for
  bool last_deferred_solve = false;
  const float *d_ll_override = NULL;             // kamd_batch_decoder_set_loglike_override
  hipStream_t s_main = NULL;
  hipEvent_t ev[3] = {};
  // A shard so small that its search is the longest utterance's own chain of frames (8 ranks over test-clean): the
  // `long_lanes` longest utterances are scored first and searched by a second decoder object on its own stream while
  // the acoustic model of the rest is still running (kamd_batch_decoder_set_long_decoder).
  kamd_decoder *dec_long = NULL;
  int long_lanes = 0;
  hipStream_t s_long = NULL;
  hipEvent_t ev_long = NULL;
  std::vector<int64_t> ll_row;          // first row of utterance k's log-likelihoods in d_ll (== out_off unless split)
  std::vector<int> map_long, map_rest;  // queue-local utterance number -> position in `kept`
  std::vector<int> map_retry;           // ... of a second-chance launch (kamd_decoder_queue_launch_wide)
  std::vector<const float *> task_ll;   // per position in `kept`: the first log-likelihood row its task was given
  bool last_split = false;
  std::vector<int64_t> load_row;        // load_host: utterance k's first row in a matrix whose rows follow the LOAD order (kamd_batch_decoder_set_loglike_override)
  int host_split = 0;                   // load_host put this many of the longest utterances first (pass 0): they go to dec_long
  std::vector<UttOut> out;
  // host-tail pool
  std::vector<std::thread> workers;
  std::vector<hipStream_t> copy_streams;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<int> jobs;
  int pending = 0;
  bool quit = false;
  kamd_batch_stats last = {};
};

static void Unregister(BatchDecoder *b) {
  b->host_unloaded = false;
  if (b->h_registered) { (void)hipHostUnregister(b->h_registered); b->h_registered = NULL; }
}

template <typename T>
static int GrowDev(T **p, size_t *cap, size_t need) {
  if (need <= *cap) return KAMD_OK;
  if (*p) (void)hipFree(*p);
  *p = NULL; *cap = 0;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(p), need * sizeof(T)));
  *cap = need;
  return KAMD_OK;
}

// One finished utterance: what DecodeUtteranceLatticeFaster does after Decode()
// (decoder/decoder-wrappers.cc:217-296): best path -> words / alignment / weight, raw lattice,
// optional DeterminizeLatticePhonePrunedWrapper.
#define KAMD_JOB_LONG (1 << 30)     // job = queue-local utterance number, | this bit for the long utterances' queue
#define KAMD_JOB_RETRY (1 << 29)    // ... | this bit for a second-chance launch on the main decoder (map_retry)
static void HostTail(BatchDecoder *b, int job, hipStream_t cs) {
  const auto t0 = std::chrono::steady_clock::now();
  const bool lq = (job & KAMD_JOB_LONG) != 0, rq = (job & KAMD_JOB_RETRY) != 0;
  const int u = job & ~(KAMD_JOB_LONG | KAMD_JOB_RETRY);               // the queue's utterance number
  kamd_decoder *dec = lq ? b->dec_long : b->dec;
  const int k = rq ? b->map_retry[u] : lq ? b->map_long[u] : (b->last_split ? b->map_rest[u] : u);   // position in `kept`
  UttOut &o = b->out[b->kept[k]];
  o.Clear();                             // what the previous run left in this slot
  int rc = kamd_decoder_queue_result(dec, u, &o.rec);
  if (rc == KAMD_OK)
    rc = kamd_decoder_queue_fetch_lattice(dec, u, cs, &o.num_states, &o.num_arcs, &o.start, &o.st_frame, &o.st_hclg, &o.st_cost,
                                          &o.st_final, &o.arcs);
  if (rc == KAMD_OK && o.num_states > 0) {
    const int cap = std::max(o.num_arcs, 1);
    o.words.resize(cap); o.ali.resize(cap);
    int na = 0, nw = 0;
    const int brc = kamd_lattice_best_path(o.num_states, o.start, o.st_final, o.arcs, o.num_arcs, o.ali.data(), cap, &na, o.words.data(), cap,
                                           &nw, &o.graph_cost, &o.ac_cost);
    if (brc == KAMD_OK) { o.have_path = true; o.ali.resize(na); o.words.resize(nw); }
    else { o.ali.clear(); o.words.clear(); }
    if (b->opts.determinize && o.have_path) {
      std::vector<float> fin2(2 * static_cast<size_t>(o.num_states), INFINITY);
      for (int s = 0; s < o.num_states; s++)
        if (o.st_final[s] != INFINITY) { fin2[2 * s] = o.st_final[s]; fin2[2 * s + 1] = 0.0f; }
      kamd_determinize_opts dopt = b->opts.det;
      const bool phones = !b->tid_phone.empty();
      if (!phones) dopt.phone_determinize = 0;
      o.clat = kamd_lattice_determinize_phone_pruned(o.num_states, o.start, fin2.data(), o.arcs, o.num_arcs,
                                                     phones ? b->tid_phone.data() : NULL, phones ? static_cast<int>(b->tid_phone.size()) - 1 : 0,
                                                     b->opts.lattice_beam, &dopt);
      if (!o.clat) rc = KAMD_ERR_STATE;
    }
    if (!b->opts.keep_raw_lattices) {
      free(o.st_frame); free(o.st_hclg); free(o.st_cost); free(o.st_final); free(o.arcs);
      o.st_frame = o.st_hclg = NULL; o.st_cost = o.st_final = NULL; o.arcs = NULL;
    }
  }
  if (rc != KAMD_OK) { o.status = rc; o.message = LastError(); }
  o.host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  o.done = 1;
}

static void WorkerLoop(BatchDecoder *b, int idx) {
  (void)hipSetDevice(b->device);
  hipStream_t cs = b->copy_streams[idx];
  for (;;) {
    int u;
    {
      std::unique_lock<std::mutex> lk(b->mu);
      b->cv_job.wait(lk, [&] { return b->quit || !b->jobs.empty(); });
      if (b->jobs.empty()) return;   // quit
      u = b->jobs.front(); b->jobs.pop_front();
    }
    HostTail(b, u, cs);
    {
      std::lock_guard<std::mutex> lk(b->mu);
      b->pending--;
    }
    b->cv_done.notify_all();
  }
}

}  // namespace kamd
using kamd::BatchDecoder;

// Do the longest utterances go to the second decoder object?  Only when the longest utterance's chain of frames, at the
// per-frame cost of a lane on an otherwise idle device (~0.63 of a loaded lane's: 79 against 125 us at the matched load),
// is about as long as the balanced share of the whole shard or longer.  Measured on the 2620-utterance set: a shard of 8
// goes 132.8 -> 109 ms (32 lanes), a shard of 4 182.3 -> 174.0 ms, a shard of 2 loses (308 -> 333 ms: the model beside
// the long lanes runs 1.2x slower and there is no chain to hide).  KAMD_BATCH_SPLIT=0/1 overrides.
static bool WantSplit(const BatchDecoder *b, int n, int64_t longest_out, int64_t total_out) {
  if (!b->dec_long || b->long_lanes <= 0) return false;
  if (const char *e = getenv("KAMD_BATCH_SPLIT")) return n > b->long_lanes && atoi(e) != 0;
  const int lanes_main = b->opts.resident_lanes > 0 ? b->opts.resident_lanes : kamd_device_num_cus() * kamd_decoder_lanes_per_cu();
  return n >= 4 * b->long_lanes && 0.7 * static_cast<double>(longest_out) > static_cast<double>(total_out) / std::max(lanes_main, 1);
}

extern "C" {

void kamd_batch_opts_default(kamd_batch_opts *o) {
  memset(o, 0, sizeof(*o));
  o->resident_lanes = 0; o->host_threads = 8; o->determinize = 1; o->keep_raw_lattices = 0;
  o->nnet_pass_frames = 1000000; o->lattice_pool_bytes = 1ll << 30; o->lattice_beam = 8.0f;
  kamd_determinize_opts_default(&o->det);
  o->first_pass_frames = 60000;
}

kamd_batch_decoder *kamd_batch_decoder_create(kamd_feat *feat, kamd_nnet *nnet, kamd_decoder *dec, const kamd_batch_opts *opts,
                                              const int32_t *tid_phone, int32_t num_tids) {
  if (!nnet || !dec) { kamd::SetError(KAMD_ERR_ARG, "batch decoder: null stage handle"); return NULL; }
  if (!kamd::RequireDevice()) return NULL;
  BatchDecoder *b = new BatchDecoder();
  b->feat = feat; b->nnet = nnet; b->dec = dec;
  if (opts) b->opts = *opts; else kamd_batch_opts_default(&b->opts);
  if (b->opts.host_threads < 1) b->opts.host_threads = 1;
  if (b->opts.host_threads > 256) b->opts.host_threads = 256;
  if (b->opts.nnet_pass_frames <= 0) b->opts.nnet_pass_frames = 1000000;
  if (tid_phone && num_tids > 0) b->tid_phone.assign(tid_phone, tid_phone + num_tids + 1);
  b->feat_dim = feat ? kamd_feat_dim(feat) : kamd_nnet_input_dim(nnet);
  if (b->feat_dim != kamd_nnet_input_dim(nnet)) {
    kamd::SetError(KAMD_ERR_ARG, "batch decoder: features have dim %d, the model's input node %d", b->feat_dim, kamd_nnet_input_dim(nnet));
    delete b;
    return NULL;
  }
  b->ld_feat = kamd::RoundUp(b->feat_dim, 16); b->P = kamd_nnet_output_dim(nnet);
  bool ok = hipGetDevice(&b->device) == hipSuccess && hipStreamCreateWithFlags(&b->s_main, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < 3; i++) ok = hipEventCreate(&b->ev[i]) == hipSuccess;
  b->copy_streams.resize(b->opts.host_threads, NULL);
  for (int i = 0; ok && i < b->opts.host_threads; i++) ok = hipStreamCreateWithFlags(&b->copy_streams[i], hipStreamNonBlocking) == hipSuccess;
  if (ok && b->opts.lattice_pool_bytes > 0) ok = kamd_decoder_queue_configure(dec, b->opts.lattice_pool_bytes) == KAMD_OK;
  if (!ok) {
    kamd::SetError(KAMD_ERR_HIP, "batch decoder: stream / event creation failed");
    kamd_batch_decoder_destroy(reinterpret_cast<kamd_batch_decoder *>(b));
    return NULL;
  }
  for (int i = 0; i < b->opts.host_threads; i++) b->workers.emplace_back(kamd::WorkerLoop, b, i);
  return reinterpret_cast<kamd_batch_decoder *>(b);
}

void kamd_batch_decoder_destroy(kamd_batch_decoder *h) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (!b) return;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    b->quit = true;
  }
  b->cv_job.notify_all();
  for (std::thread &t : b->workers) t.join();
  for (kamd::UttOut &o : b->out) o.Clear();
  for (hipStream_t s : b->copy_streams) if (s) (void)hipStreamDestroy(s);
  for (int i = 0; i < 3; i++) if (b->ev[i]) (void)hipEventDestroy(b->ev[i]);
  if (b->s_main) (void)hipStreamDestroy(b->s_main);
  if (b->s_long) (void)hipStreamDestroy(b->s_long);
  if (b->ev_long) (void)hipEventDestroy(b->ev_long);
  for (int i = 0; i < 2; i++) if (b->ev_solve[i]) (void)hipEventDestroy(b->ev_solve[i]);
  if (b->d_waves) (void)hipFree(b->d_waves);
  if (b->d_feats) (void)hipFree(b->d_feats);
  if (b->d_ll) (void)hipFree(b->d_ll);
  if (b->d_iv) (void)hipFree(b->d_iv);
  if (b->d_feat_meta) (void)hipFree(b->d_feat_meta);
  kamd::Unregister(b);
  for (int i = 0; i < BatchDecoder::kStageBufs; i++) {
    if (b->h_stage[i]) (void)hipHostFree(b->h_stage[i]);
    if (b->ev_stage[i]) (void)hipEventDestroy(b->ev_stage[i]);
  }
  for (std::vector<hipEvent_t> *v : {&b->ev_up, &b->ev_f0, &b->ev_f1, &b->ev_n1, &b->ev_iv0, &b->ev_iv1, &b->ev_n0})
    for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
  if (b->d_oiv) (void)hipFree(b->d_oiv);
  if (b->s_up) (void)hipStreamDestroy(b->s_up);
  delete b;
}

namespace kamd {
// the model's ivector input: fed by the extractor (rows of d_oiv), or the set must come through load_features
static int CheckIvectorInput(const BatchDecoder *b) {
  if (kamd_nnet_ivector_dim(b->nnet) > 0 && !b->iv_extractor)
    return SetError(KAMD_ERR_ARG, "the model has an ivector input: set an extractor (kamd_batch_decoder_set_ivector_extractor) or use "
                    "kamd_batch_decoder_load_features");
  return KAMD_OK;
}
static int SizeOnlineIvectors(BatchDecoder *b) {
  b->oiv_off.assign(1, 0);
  if (!b->iv_extractor || b->have_iv) return KAMD_OK;
  for (size_t k = 0; k + 1 < b->feat_off.size(); k++)
    b->oiv_off.push_back(b->oiv_off.back() + kamd_ivector_num_ivectors(b->iv_extractor, static_cast<int>(b->feat_off[k + 1] - b->feat_off[k])));
  return GrowDev(&b->d_oiv, &b->oiv_cap, std::max<size_t>(static_cast<size_t>(b->oiv_off.back()) * kamd_ivector_dim(b->iv_extractor), 1));
}
}  // namespace kamd

int kamd_batch_decoder_set_ivector_extractor(kamd_batch_decoder *h, kamd_ivector_extractor *e, int frames_per_chunk) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (!e) { b->iv_extractor = NULL; return KAMD_OK; }
  if (frames_per_chunk <= 0) return kamd::SetError(KAMD_ERR_ARG, "bad frames per chunk");
  if (kamd_ivector_dim(e) != kamd_nnet_ivector_dim(b->nnet))
    return kamd::SetError(KAMD_ERR_ARG, "the extractor gives %d-dim ivectors, the model takes %d", kamd_ivector_dim(e), kamd_nnet_ivector_dim(b->nnet));
  if (b->n_utts > 0) return kamd::SetError(KAMD_ERR_STATE, "set the extractor before the test set is loaded");
  b->iv_extractor = e; b->frames_per_chunk = frames_per_chunk;
  return KAMD_OK;
}

int kamd_batch_decoder_set_chunk_rule(kamd_batch_decoder *h, int rule) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (rule != 0 && rule != 1) return kamd::SetError(KAMD_ERR_ARG, "chunk rule: 0 (DecodableNnetSimple) or 1 (NnetBatchComputer)");
  b->chunk_rule = rule;
  return KAMD_OK;
}

int kamd_batch_decoder_load(kamd_batch_decoder *h, const float *waves, const int64_t *h_wave_off, int n_utts) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (n_utts <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty test set");
  if (!b->feat) return kamd::SetError(KAMD_ERR_STATE, "batch decoder was created without a feature stage: use kamd_batch_decoder_load_features");
  if (kamd::CheckIvectorInput(b) != KAMD_OK) return KAMD_ERR_ARG;
  for (kamd::UttOut &o : b->out) o.Clear();
  b->n_utts = 0; b->from_features = false; b->have_iv = false; b->h_waves = NULL;
  kamd::Unregister(b);
  b->kept.clear(); b->skipped.clear();
  b->wave_off.assign(1, 0); b->feat_off.assign(1, 0); b->out_off.assign(1, 0);
  for (int u = 0; u < n_utts; u++) {
    const int64_t len = h_wave_off[u + 1] - h_wave_off[u];
    const int T = len > 0 ? kamd_feat_num_frames(b->feat, len) : 0;
    if (len < 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d: negative length", u);
    if (T <= 0) { b->skipped.push_back(u); continue; }
    b->kept.push_back(u);
    b->wave_off.push_back(b->wave_off.back() + len);
    b->feat_off.push_back(b->feat_off.back() + T);
    b->out_off.push_back(b->out_off.back() + kamd_nnet_num_output_frames(b->nnet, T));
  }
  const size_t ns = static_cast<size_t>(b->wave_off.back());
  if (kamd::GrowDev(&b->d_waves, &b->waves_cap, std::max<size_t>(ns, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&b->d_feats, &b->feats_cap, std::max<size_t>(static_cast<size_t>(b->feat_off.back()) * b->ld_feat, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&b->d_ll, &b->ll_cap, std::max<size_t>(static_cast<size_t>(b->out_off.back()) * b->P, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::SizeOnlineIvectors(b) != KAMD_OK) return KAMD_ERR_HIP;
  if (b->skipped.empty()) {
    if (ns) KAMD_HIP(hipMemcpy(b->d_waves, waves + h_wave_off[0], ns * sizeof(float), hipMemcpyHostToDevice));
  } else {
    for (size_t k = 0; k < b->kept.size(); k++) {
      const int u = b->kept[k];
      KAMD_HIP(hipMemcpy(b->d_waves + b->wave_off[k], waves + h_wave_off[u], static_cast<size_t>(h_wave_off[u + 1] - h_wave_off[u]) * sizeof(float),
                         hipMemcpyHostToDevice));
    }
  }
  b->out.resize(n_utts);
  b->n_utts = n_utts;
  return KAMD_OK;
}

int kamd_batch_decoder_load_features(kamd_batch_decoder *h, const float *feats, const int64_t *row_off, int dim, const float *ivectors,
                                     int ivector_dim, int n_utts) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (n_utts <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty test set");
  if (dim != b->feat_dim) return kamd::SetError(KAMD_ERR_ARG, "features have dim %d, the model's input node %d", dim, b->feat_dim);
  // the model's ivector input: one vector per utterance (--ivectors), or -- with an extractor set and none given -- online
  // i-vectors estimated from these very features (--online-ivectors of the recipe)
  const bool online = b->iv_extractor != NULL && ivectors == NULL;
  const int want_iv = online ? 0 : kamd_nnet_ivector_dim(b->nnet);
  if (!online && (ivectors ? ivector_dim : 0) != want_iv)
    return kamd::SetError(KAMD_ERR_ARG, "model expects ivector dim %d, got %d", want_iv, ivectors ? ivector_dim : 0);
  for (kamd::UttOut &o : b->out) o.Clear();
  b->n_utts = 0; b->h_waves = NULL;
  kamd::Unregister(b);
  b->kept.clear(); b->skipped.clear();
  b->feat_off.assign(1, 0); b->out_off.assign(1, 0);
  for (int u = 0; u < n_utts; u++) {
    const int64_t T = row_off[u + 1] - row_off[u];
    if (T < 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d: negative length", u);
    if (T == 0) { b->skipped.push_back(u); continue; }        // "Zero-length utterance", nnet3-latgen-faster-batch.cc:184
    b->kept.push_back(u);
    b->feat_off.push_back(b->feat_off.back() + T);
    b->out_off.push_back(b->out_off.back() + kamd_nnet_num_output_frames(b->nnet, static_cast<int>(T)));
  }
  const size_t rows = static_cast<size_t>(b->feat_off.back());       // rows of empty utterances: none, so the rest stay contiguous
  if (kamd::GrowDev(&b->d_feats, &b->feats_cap, std::max<size_t>(rows * b->ld_feat, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&b->d_ll, &b->ll_cap, std::max<size_t>(static_cast<size_t>(b->out_off.back()) * b->P, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (rows) {
    KAMD_HIP(hipMemset(b->d_feats, 0, rows * b->ld_feat * sizeof(float)));
    KAMD_HIP(hipMemcpy2D(b->d_feats, b->ld_feat * sizeof(float), feats + static_cast<size_t>(row_off[0]) * dim, dim * sizeof(float),
                         dim * sizeof(float), rows, hipMemcpyHostToDevice));
  }
  if (want_iv > 0 && !b->kept.empty()) {
    std::vector<float> iv(b->kept.size() * static_cast<size_t>(want_iv));
    for (size_t k = 0; k < b->kept.size(); k++)
      memcpy(&iv[k * want_iv], ivectors + static_cast<size_t>(b->kept[k]) * want_iv, sizeof(float) * want_iv);
    if (kamd::GrowDev(&b->d_iv, &b->iv_cap, iv.size()) != KAMD_OK) return KAMD_ERR_HIP;
    KAMD_HIP(hipMemcpy(b->d_iv, iv.data(), iv.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  b->from_features = true; b->have_iv = want_iv > 0;
  if (kamd::SizeOnlineIvectors(b) != KAMD_OK) return KAMD_ERR_HIP;
  b->out.resize(n_utts);
  b->n_utts = n_utts;
  return KAMD_OK;
}

int kamd_batch_decoder_unload_host(kamd_batch_decoder *h) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (!b) return kamd::SetError(KAMD_ERR_ARG, "null batch decoder");
  kamd::Unregister(b);
  if (b->h_waves) { b->h_waves = NULL; b->host_unloaded = true; }
  return KAMD_OK;
}

int kamd_batch_decoder_load_host(kamd_batch_decoder *h, const float *waves, const int64_t *h_wave_off, int n_utts) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (n_utts <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty test set");
  if (!b->feat) return kamd::SetError(KAMD_ERR_STATE, "batch decoder was created without a feature stage: use kamd_batch_decoder_load_features");
  if (kamd::CheckIvectorInput(b) != KAMD_OK) return KAMD_ERR_ARG;
  for (kamd::UttOut &o : b->out) o.Clear();
  b->n_utts = 0; b->from_features = false; b->have_iv = false; b->h_waves = NULL;
  kamd::Unregister(b);
  b->kept.clear(); b->skipped.clear(); b->h_wave_src.clear();
  b->wave_off.assign(1, 0); b->feat_off.assign(1, 0); b->out_off.assign(1, 0);
  b->host_split = 0;
  std::vector<int> order;                      // the utterances that have frames, in the order they are stored and scored
  std::vector<int> frames_of(n_utts, 0);
  int64_t longest_out = 0, total_out = 0;
  for (int u = 0; u < n_utts; u++) {
    const int64_t len = h_wave_off[u + 1] - h_wave_off[u];
    if (len < 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d: negative length", u);
    const int T = len > 0 ? kamd_feat_num_frames(b->feat, len) : 0;
    if (T <= 0) { b->skipped.push_back(u); continue; }
    frames_of[u] = T;
    order.push_back(u);
    const int64_t no = kamd_nnet_num_output_frames(b->nnet, T);
    longest_out = std::max(longest_out, no); total_out += no;
  }
  // kamd_batch_decoder_set_long_decoder: the longest utterances first, as a pass of their own -- their search starts on
  // the second decoder object as soon as that pass is scored, beside the model of the others (RunImpl)
  // (round 4: also with an extractor -- the long utterances' pass gets its own online i-vectors and chunked forward like any
  // other pass -- and with a log-likelihood override, whose rows follow load_row)
  if ((b->iv_extractor || kamd_nnet_ivector_dim(b->nnet) == 0) && WantSplit(b, static_cast<int>(order.size()), longest_out, total_out)) {
    std::vector<int> by_len(order);
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int c) { return frames_of[a] > frames_of[c]; });
    by_len.resize(b->long_lanes);
    std::vector<char> is_long(n_utts, 0);
    for (int u : by_len) is_long[u] = 1;
    std::vector<int> rest;
    for (int u : order) if (!is_long[u]) rest.push_back(u);
    order = by_len;
    order.insert(order.end(), rest.begin(), rest.end());
    b->host_split = b->long_lanes;
  }
  {
    std::vector<int64_t> row_of_utt(n_utts, 0);
    int64_t row = 0;
    for (int u = 0; u < n_utts; u++) { row_of_utt[u] = row; if (frames_of[u] > 0) row += kamd_nnet_num_output_frames(b->nnet, frames_of[u]); }
    b->load_row.clear();
    for (int u : order) b->load_row.push_back(row_of_utt[u]);
  }
  for (int u : order) {
    const int64_t len = h_wave_off[u + 1] - h_wave_off[u];
    const int T = frames_of[u];
    b->kept.push_back(u);
    b->h_wave_src.push_back(h_wave_off[u]);
    b->wave_off.push_back(b->wave_off.back() + len);
    b->feat_off.push_back(b->feat_off.back() + T);
    b->out_off.push_back(b->out_off.back() + kamd_nnet_num_output_frames(b->nnet, T));
  }
  const int n = static_cast<int>(b->kept.size());
  if (kamd::GrowDev(&b->d_waves, &b->waves_cap, std::max<size_t>(static_cast<size_t>(b->wave_off.back()), 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&b->d_feats, &b->feats_cap, std::max<size_t>(static_cast<size_t>(b->feat_off.back()) * b->ld_feat, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&b->d_ll, &b->ll_cap, std::max<size_t>(static_cast<size_t>(b->out_off.back()) * b->P, 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::SizeOnlineIvectors(b) != KAMD_OK) return KAMD_ERR_HIP;
  // passes: a small first one, so that the acoustic model starts behind a short upload
  b->pass_u0.assign(1, 0);
  if (b->host_split > 0) b->pass_u0.push_back(b->host_split);        // (the long utterances ARE the small first pass)
  for (int u0 = b->pass_u0.back(); u0 < n;) {
    const int64_t cap = (b->pass_u0.size() == 1 && b->opts.first_pass_frames > 0) ? b->opts.first_pass_frames : b->opts.nnet_pass_frames;
    int u1 = u0 + 1;
    while (u1 < n && b->feat_off[u1 + 1] - b->feat_off[u0] <= cap) u1++;
    b->pass_u0.push_back(u1);
    u0 = u1;
  }
  const int np = static_cast<int>(b->pass_u0.size()) - 1;
  // feature offsets of every pass, uploaded once
  std::vector<int64_t> meta;
  b->feat_meta_off.assign(np, 0); b->pass_frames.assign(np, 0);
  for (int p = 0; p < np; p++) {
    const int u0 = b->pass_u0[p], cnt = b->pass_u0[p + 1] - u0;
    b->feat_meta_off[p] = meta.size();
    meta.resize(meta.size() + 3 * static_cast<size_t>(cnt + 1));
    kamd::FeatBuildMeta(b->feat, b->wave_off.data() + u0, cnt, b->feat_off.data() + u0, meta.data() + b->feat_meta_off[p], &b->pass_frames[p]);
  }
  if (kamd::GrowDev(&b->d_feat_meta, &b->feat_meta_cap, std::max<size_t>(meta.size(), 1)) != KAMD_OK) return KAMD_ERR_HIP;
  if (!meta.empty()) KAMD_HIP(hipMemcpy(b->d_feat_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice));
  if (!b->s_up) KAMD_HIP(hipStreamCreateWithFlags(&b->s_up, hipStreamNonBlocking));
  for (int i = 0; i < BatchDecoder::kStageBufs; i++) {
    if (!b->h_stage[i]) KAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&b->h_stage[i]), BatchDecoder::kStageFloats * sizeof(float), hipHostMallocDefault));
    if (!b->ev_stage[i]) KAMD_HIP(hipEventCreateWithFlags(&b->ev_stage[i], hipEventDisableTiming));
  }
  for (std::vector<hipEvent_t> *v : {&b->ev_up, &b->ev_f0, &b->ev_f1, &b->ev_n1})
    while (static_cast<int>(v->size()) < np) {
      hipEvent_t e = NULL;
      KAMD_HIP(hipEventCreate(&e));
      v->push_back(e);
    }
  // page-lock the caller's samples where they are (once, here): the copies of run() then need no staging pass.  A runtime
  // that refuses (limits on locked memory) leaves the staged path.
  static const bool no_register = getenv("KAMD_BATCH_NO_REGISTER") != NULL;
  const size_t span = static_cast<size_t>(h_wave_off[n_utts] - h_wave_off[0]) * sizeof(float);
  b->reg_lo = b->reg_hi = 0;
  if (!no_register && span > 0) {
    // Only the pages that lie WHOLLY inside the caller's samples are locked.  The runtime locks whole pages: had the range
    // been handed over as it is, the first and the last page would be shared with whatever the caller's allocator put next
    // to the array, and to the runtime those neighbours would look like page-locked memory from then on -- a copy of the
    // host's own (torch, a Kaldi binary) from such a neighbour is then started as a pinned copy and runs off the end of what
    // is pinned.  The head and the tail of the array (under a page each) go through the bounce buffer instead.
    const uintptr_t page = static_cast<uintptr_t>(sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096);
    const uintptr_t lo = reinterpret_cast<uintptr_t>(waves + h_wave_off[0]), hi = lo + span;
    const uintptr_t in_lo = (lo + page - 1) / page * page, in_hi = hi / page * page;
    if (in_hi > in_lo && hipHostRegister(reinterpret_cast<void *>(in_lo), in_hi - in_lo, hipHostRegisterDefault) == hipSuccess) {
      b->h_registered = reinterpret_cast<void *>(in_lo);
      b->reg_lo = in_lo; b->reg_hi = in_hi;
    } else {
      (void)hipGetLastError();
    }
  }
  b->h_waves = waves;
  b->out.resize(n_utts);
  b->n_utts = n_utts;
  return KAMD_OK;
}

int kamd_batch_decoder_set_loglike_override(kamd_batch_decoder *h, const float *d_loglikes) {
  reinterpret_cast<BatchDecoder *>(h)->d_ll_override = d_loglikes;
  return KAMD_OK;
}

int64_t kamd_batch_decoder_output_frames(kamd_batch_decoder *h, int32_t *frames, int cap) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (frames) for (int u = 0; u < std::min(cap, b->n_utts); u++) frames[u] = 0;
  if (frames)
    for (size_t k = 0; k < b->kept.size(); k++)
      if (b->kept[k] < cap) frames[b->kept[k]] = static_cast<int32_t>(b->out_off[k + 1] - b->out_off[k]);
  return b->out_off.empty() ? 0 : b->out_off.back();
}

int kamd_batch_decoder_set_long_decoder(kamd_batch_decoder *h, kamd_decoder *dec_long, int lanes) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (!dec_long || lanes <= 0) { b->dec_long = NULL; b->long_lanes = 0; return KAMD_OK; }
  if (!b->s_long) {
    // a stream of the highest priority: HIP shares a few hardware queues between the streams of one priority, and two
    // streams on one queue run their kernels in submission order -- seen in a rocprofv3 trace: the model's GEMMs started
    // only when the long utterances' search, issued before them on "another" stream, had ended.  Priorities have queues
    // of their own.  (And the chain of the longest utterance IS the critical path of a small shard.)
    int least = 0, greatest = 0;
    KAMD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    KAMD_HIP(hipStreamCreateWithPriority(&b->s_long, hipStreamNonBlocking, greatest));
  }
  if (!b->ev_long) KAMD_HIP(hipEventCreateWithFlags(&b->ev_long, hipEventDisableTiming));
  if (b->opts.lattice_pool_bytes > 0 && kamd_decoder_queue_configure(dec_long, std::max<int64_t>(b->opts.lattice_pool_bytes / 8, 1 << 24)) != KAMD_OK)
    return KAMD_ERR_HIP;
  b->dec_long = dec_long; b->long_lanes = lanes;
  return KAMD_OK;
}

static int RunImpl(BatchDecoder *b, kamd_batch_stats *stats);


// Every error exit of a run leaves through here: jobs may be queued to the host-tail pool and a work-queue kernel may
// still be running; nothing of this object may be reused (load(), run()) before both have ended.
int kamd_batch_decoder_run(kamd_batch_decoder *h, kamd_batch_stats *stats) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (b->host_unloaded) return kamd::SetError(KAMD_ERR_STATE, "kamd_batch_decoder_run: the waveforms were released (kamd_batch_decoder_unload_host); load again");
  const int rc = RunImpl(b, stats);
  if (rc != KAMD_OK) {
    {
      std::unique_lock<std::mutex> lk(b->mu);
      b->cv_done.wait(lk, [&] { return b->pending == 0; });
    }
    if (b->s_main) (void)hipStreamSynchronize(b->s_main);
    if (b->s_long) (void)hipStreamSynchronize(b->s_long);
    if (b->s_up) (void)hipStreamSynchronize(b->s_up);
    (void)hipGetLastError();          // (the error that is reported is the one RunImpl set)
  }
  return rc;
}

static int RunImpl(BatchDecoder *b, kamd_batch_stats *stats) {
  if (b->n_utts <= 0) return kamd::SetError(KAMD_ERR_STATE, "no test set loaded");
  const int n = static_cast<int>(b->kept.size());
  // the previous run's lattices are freed by the worker that fills the slot again (HostTail): 2620 compact lattices are
  // milliseconds of free() that need not sit in front of the first launch
  for (kamd::UttOut &o : b->out) { o.done = 0; o.status = KAMD_OK; }
  for (int u : b->skipped) {
    kamd::UttOut &o = b->out[u];
    o.Clear();
    memset(&o.rec, 0, sizeof(o.rec));
    o.status = KAMD_ERR_ARG; o.message = "utterance " + std::to_string(u) + ": too short for one frame"; o.done = 1;
  }
  if (n == 0) {
    kamd_batch_stats s;
    memset(&s, 0, sizeof(s));
    s.n_failed = static_cast<int32_t>(b->skipped.size());
    b->last = s;
    if (stats) *stats = s;
    return KAMD_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  static const bool trace = getenv("KAMD_BATCH_TRACE") != NULL;          // host-side timeline of a run on stderr
  auto mark = [&](const char *what, int k) {
    if (trace) fprintf(stderr, "[batch %8.2f ms] %s %d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what, k);
  };
  hipStream_t st = b->s_main;
  KAMD_HIP(hipEventRecord(b->ev[0], st));
  int rc = KAMD_OK;
  const bool host_mode = b->h_waves != NULL;
  // ---- kamd_batch_decoder_load_host: one thread stages the caller's samples through page-locked buffers and issues the
  // copies pass by pass on the copy stream; this thread launches a pass's features and model behind its event
  struct Upload {
    std::mutex mu; std::condition_variable cv;
    int issued = 0, rc = KAMD_OK; std::string err; double done_ms = 0;
  } up;
  std::thread uploader;
  if (host_mode) {
    uploader = std::thread([b, &up, t0]() {
      (void)hipSetDevice(b->device);
      const int np = static_cast<int>(b->pass_u0.size()) - 1;
      int k = 0;
      auto fail = [&](hipError_t e) {
        std::lock_guard<std::mutex> lk(up.mu);
        up.rc = KAMD_ERR_HIP; up.err = std::string("waveform upload failed: ") + hipGetErrorString(e);
      };
      for (int p = 0; p < np && up.rc == KAMD_OK; p++) {
        for (int u = b->pass_u0[p]; u < b->pass_u0[p + 1] && up.rc == KAMD_OK;) {
          // the longest run of utterances that is contiguous in the caller's buffer (all of them unless some were skipped)
          int v = u + 1;
          while (v < b->pass_u0[p + 1] && b->h_wave_src[v] == b->h_wave_src[v - 1] + (b->wave_off[v] - b->wave_off[v - 1])) v++;
          const float *src = b->h_waves + b->h_wave_src[u];
          float *dst = b->d_waves + b->wave_off[u];
          if (b->h_registered) {          // page-locked in place: one copy for the whole run of utterances
            // ... cut at the ends of the locked pages: the piece inside is read where it lies, the part of the array's first
            // and last page outside goes through the bounce buffer (MemcpyAsyncSafe finds it pageable)
            const uintptr_t s0 = reinterpret_cast<uintptr_t>(src), s1 = s0 + static_cast<size_t>(b->wave_off[v] - b->wave_off[u]) * sizeof(float);
            const uintptr_t cut[4] = {s0, std::min(std::max(s0, b->reg_lo), s1), std::min(std::max(s0, b->reg_hi), s1), s1};
            for (int i = 0; i < 3; i++)
              if (cut[i + 1] > cut[i]) {
                const hipError_t e = hipMemcpyAsync(reinterpret_cast<char *>(dst) + (cut[i] - s0), reinterpret_cast<const void *>(cut[i]), cut[i + 1] - cut[i],
                                                    hipMemcpyHostToDevice, b->s_up);
                if (e != hipSuccess) { fail(e); break; }
              }
            u = v;
            continue;
          }
          for (int64_t left = b->wave_off[v] - b->wave_off[u]; left > 0;) {
            const size_t cnt = static_cast<size_t>(std::min<int64_t>(left, BatchDecoder::kStageFloats));
            const int i = k % BatchDecoder::kStageBufs;
            hipError_t e = k >= BatchDecoder::kStageBufs ? hipEventSynchronize(b->ev_stage[i]) : hipSuccess;
            if (e == hipSuccess) {
              memcpy(b->h_stage[i], src, cnt * sizeof(float));
              e = hipMemcpyAsync(dst, b->h_stage[i], cnt * sizeof(float), hipMemcpyHostToDevice, b->s_up);
            }
            if (e == hipSuccess) e = hipEventRecord(b->ev_stage[i], b->s_up);
            if (e != hipSuccess) { fail(e); break; }
            src += cnt; dst += cnt; left -= static_cast<int64_t>(cnt); k++;
          }
          u = v;
        }
        // the pass is handed to the launch thread once its last byte is in HBM, by this thread's own wait: a
        // hipStreamWaitEvent on the main stream against an event of the copy stream held the kernels back until EVERY copy
        // queued by then had finished (seen in the rocprofv3 timeline: pass 0's features started behind the last pass's
        // copy, 21 ms late)
        if (up.rc == KAMD_OK) {
          const hipError_t e = hipStreamSynchronize(b->s_up);
          if (e != hipSuccess) fail(e);
        }
        {
          std::lock_guard<std::mutex> lk(up.mu);
          up.issued = p + 1;
        }
        up.cv.notify_all();
      }
      if (up.rc == KAMD_OK) {
        const hipError_t e = hipStreamSynchronize(b->s_up);
        if (e != hipSuccess) fail(e);
      }
      std::lock_guard<std::mutex> lk(up.mu);
      up.issued = np;
      up.done_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      up.cv.notify_all();
    });
  }
  // whatever happens below, the uploader is joined before this function returns (it reads the caller's buffer)
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};
  if (!b->from_features && !host_mode) rc = kamd_feat_compute_batch_device(b->feat, b->d_waves, b->wave_off.data(), n, b->d_feats, b->feat_off.data(), b->ld_feat, st);
  if (rc != KAMD_OK) return rc;
  KAMD_HIP(hipEventRecord(b->ev[1], st));
  double upload_wait_ms = 0, first_pass_start_ms = 0;
  // ---- acoustic model, a few passes of <= nnet_pass_frames input frames
  double flops = 0;
  int passes = 0;
  b->ll_row.assign(b->out_off.begin(), b->out_off.end() - 1);
  int64_t longest = 0;
  for (int k = 0; k < n; k++) longest = std::max(longest, b->out_off[k + 1] - b->out_off[k]);
  b->task_ll.assign(n, NULL);
  bool split = !b->have_iv && WantSplit(b, n, longest, b->out_off.back());
  if (host_mode || b->d_ll_override) split = false;       // (the passes follow the upload -- see host_split below; a planted matrix is in load order)
  const bool online_iv = b->iv_extractor != NULL && !b->have_iv;
  if (online_iv) split = false;
  const float *ll_base = b->d_ll_override ? b->d_ll_override : b->d_ll;
  // load_host stored the long utterances first: pass 0 is theirs, the second decoder object takes them from there
  const int Kh = (host_mode && b->dec_long && b->host_split == b->long_lanes && b->host_split < n) ? b->host_split : 0;
  const bool two_queues = split || Kh > 0;
  b->last_split = two_queues;
  int n_main = n;
  std::vector<kamd_queue_task> tasks;
  if (!split) {
    // the passes: load_host fixed them (pass_u0); otherwise <= nnet_pass_frames input frames each
    std::vector<int> pu(1, 0);
    if (host_mode) pu = b->pass_u0;
    else
      for (int u0 = 0; u0 < n;) {
        int u1 = u0 + 1;
        while (u1 < n && b->feat_off[u1 + 1] - b->feat_off[u0] <= b->opts.nnet_pass_frames) u1++;
        pu.push_back(u1);
        u0 = u1;
      }
    const int np = static_cast<int>(pu.size()) - 1;
    // what runs in front of a pass's acoustic model, on stream `ps`: (load_host) the pass's features behind its copies
    auto features_of = [&](int pass, hipStream_t ps) -> int {
      if (!host_mode) return KAMD_OK;
      const auto tw = std::chrono::steady_clock::now();
      {
        std::unique_lock<std::mutex> lk(up.mu);
        up.cv.wait(lk, [&] { return up.issued > pass || up.rc != KAMD_OK; });
        if (up.rc != KAMD_OK) return kamd::SetError(up.rc, "%s", up.err.c_str());
      }
      upload_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
      KAMD_HIP(hipEventRecord(b->ev_f0[pass], ps));       // (the pass's samples are in HBM: the uploader waited for them)
      const int frc = kamd::FeatLaunchPremeta(b->feat, b->d_waves, b->d_feat_meta + b->feat_meta_off[pass], pu[pass + 1] - pu[pass],
                                              b->pass_frames[pass], b->d_feats, b->ld_feat, ps);
      if (frc != KAMD_OK) return frc;
      KAMD_HIP(hipEventRecord(b->ev_f1[pass], ps));
      if (pass == 0) first_pass_start_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      mark("features issued, pass", pass);
      return KAMD_OK;
    };
    // online i-vectors: features and OnlineIvectorFeature of pass p + 1 are issued behind the acoustic model of pass p, on
    // the same stream.  (Measured on a stream of their own beside the model, 2620 utterances: the extraction's 125 ms were
    // hidden and the GEMMs took 83 ms longer -- both are throughput kernels; the wall time did not move.)
    auto ivectors_of = [&](int pass) -> int {
      while (static_cast<int>(b->ev_iv0.size()) <= pass) {
        hipEvent_t e0 = NULL, e1 = NULL, e2 = NULL;
        KAMD_HIP(hipEventCreate(&e0)); KAMD_HIP(hipEventCreate(&e1)); KAMD_HIP(hipEventCreate(&e2));
        b->ev_iv0.push_back(e0); b->ev_iv1.push_back(e1); b->ev_n0.push_back(e2);
      }
      int irc = features_of(pass, st);
      if (irc != KAMD_OK) return irc;
      KAMD_HIP(hipEventRecord(b->ev_iv0[pass], st));
      irc = kamd_ivector_extract_online_device(b->iv_extractor, b->d_feats, b->feat_off.data() + pu[pass], b->ld_feat, pu[pass + 1] - pu[pass],
                                               b->d_oiv, b->oiv_off.data() + pu[pass], st);
      if (irc != KAMD_OK) return irc;
      mark("i-vectors issued, pass", pass);
      KAMD_HIP(hipEventRecord(b->ev_iv1[pass], st));
      return KAMD_OK;
    };
    // ---- online i-vectors, the solver deferred: the statistics of every pass first (behind that pass's features, as its
    // samples arrive), then ONE SolveKernel launch for the whole set -- the solver is a sequential chain per utterance, a
    // launch costs its longest utterance's chain, and four passes paid four of them (31 ms of 98 on the 2620-utterance set,
    // against ~12 for one launch).  Not when the long utterances have a search of their own waiting for pass 0's scores, not
    // when the step statistics of the whole set (5151 doubles per i-vector row) would take more than a tenth of the free HBM.
    bool defer_solve = false;
    if (online_iv && np > 1 && Kh == 0) {
      static const bool no_defer = getenv("KAMD_IV_NO_DEFER") != NULL;
      const int I = kamd_ivector_dim(b->iv_extractor);
      const double need = static_cast<double>(b->oiv_off.back()) * (static_cast<double>(I) * (I + 1) / 2 + I + 1) * 8.0;
      size_t free_b = 0, total_b = 0;
      if (!no_defer && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (need <= 0.1 * static_cast<double>(free_b) || b->iv_reserved >= b->oiv_off.back())) {
        rc = kamd_ivector_online_reserve_steps(b->iv_extractor, b->oiv_off.back());      // (grows the buffers only when they are too small)
        if (rc != KAMD_OK) return rc;
        b->iv_reserved = std::max(b->iv_reserved, b->oiv_off.back());
        defer_solve = true;
      }
    }
    if (defer_solve) {
      for (int pass = 0; pass < np; pass++) {
        while (static_cast<int>(b->ev_iv0.size()) <= pass) {
          hipEvent_t e0 = NULL, e1 = NULL, e2 = NULL;
          KAMD_HIP(hipEventCreate(&e0)); KAMD_HIP(hipEventCreate(&e1)); KAMD_HIP(hipEventCreate(&e2));
          b->ev_iv0.push_back(e0); b->ev_iv1.push_back(e1); b->ev_n0.push_back(e2);
        }
        rc = features_of(pass, st);
        if (rc != KAMD_OK) return rc;
        KAMD_HIP(hipEventRecord(b->ev_iv0[pass], st));
        rc = kamd_ivector_online_stats_device(b->iv_extractor, b->d_feats, b->feat_off.data() + pu[pass], b->ld_feat, pu[pass + 1] - pu[pass],
                                              b->oiv_off.data() + pu[pass], st);
        if (rc != KAMD_OK) return rc;
        KAMD_HIP(hipEventRecord(b->ev_iv1[pass], st));
        mark("i-vector statistics issued, pass", pass);
      }
      if (!b->ev_solve[0]) { KAMD_HIP(hipEventCreate(&b->ev_solve[0])); KAMD_HIP(hipEventCreate(&b->ev_solve[1])); }
      KAMD_HIP(hipEventRecord(b->ev_solve[0], st));
      rc = kamd_ivector_online_solve_device(b->iv_extractor, b->feat_off.data(), n, b->d_oiv, b->oiv_off.data(), st);
      if (rc != KAMD_OK) return rc;
      KAMD_HIP(hipEventRecord(b->ev_solve[1], st));
      mark("i-vector solver issued, utterances", n);
    }
    b->last_deferred_solve = defer_solve;
    if (online_iv && !defer_solve) { rc = ivectors_of(0); if (rc != KAMD_OK) return rc; }
    for (int pass = 0; pass < np; pass++) {
      const int u0 = pu[pass], u1 = pu[pass + 1];
      if (online_iv) {
        KAMD_HIP(hipEventRecord(b->ev_n0[pass], st));
        rc = (b->chunk_rule == 1 ? kamd_nnet_forward_tasks_device : kamd_nnet_forward_chunked_device)(
            b->nnet, b->d_feats, b->feat_off.data() + u0, b->ld_feat, b->d_oiv, b->oiv_off.data() + u0, kamd_ivector_dim(b->iv_extractor),
            kamd_ivector_period(b->iv_extractor), b->frames_per_chunk, u1 - u0, b->d_ll, b->out_off.data() + u0, b->P, st);
      } else {
        rc = features_of(pass, st);
        if (rc == KAMD_OK)
          rc = kamd_nnet_forward_batch_device(b->nnet, b->d_feats, b->feat_off.data() + u0, b->ld_feat,
                                              b->have_iv ? b->d_iv + static_cast<size_t>(u0) * kamd_nnet_ivector_dim(b->nnet) : NULL, u1 - u0, b->d_ll,
                                              b->out_off.data() + u0, b->P, st);
      }
      if (rc != KAMD_OK) return rc;
      if (host_mode || online_iv) {
        while (static_cast<int>(b->ev_n1.size()) <= pass) { hipEvent_t e = NULL; KAMD_HIP(hipEventCreate(&e)); b->ev_n1.push_back(e); }
        KAMD_HIP(hipEventRecord(b->ev_n1[pass], st));
      }
      flops += kamd_nnet_last_flops(b->nnet);
      passes++;
      mark("model issued, pass", pass);
      if (pass == 0 && Kh > 0) {
        if (pu[1] != Kh) return kamd::SetError(KAMD_ERR_STATE, "the first pass is not the long utterances'");
        KAMD_HIP(hipEventRecord(b->ev_long, st));
        KAMD_HIP(hipStreamWaitEvent(b->s_long, b->ev_long, 0));
        b->map_long.resize(Kh); b->map_rest.resize(n - Kh);
        for (int k = 0; k < Kh; k++) b->map_long[k] = k;
        for (int k = Kh; k < n; k++) b->map_rest[k - Kh] = k;
        tasks.resize(Kh);
        for (int k = 0; k < Kh; k++) {                     // (stored longest first)
          const int64_t row = (b->d_ll_override && b->load_row.size() == static_cast<size_t>(n)) ? b->load_row[k] : b->out_off[k];
          tasks[k].d_loglikes = ll_base + static_cast<size_t>(row) * b->P;
          tasks[k].ld = b->P; tasks[k].n_frames = static_cast<int32_t>(b->out_off[k + 1] - b->out_off[k]);
          tasks[k].utt = k; tasks[k].reserved = 0;
          b->task_ll[k] = tasks[k].d_loglikes;
        }
        rc = kamd_decoder_queue_launch(b->dec_long, tasks.data(), Kh, Kh, b->s_long);
        if (rc != KAMD_OK) return rc;
        mark("long utterances' search issued", Kh);
      }
      if (online_iv && !defer_solve && pass + 1 < np) { rc = ivectors_of(pass + 1); if (rc != KAMD_OK) return rc; }
    }
    KAMD_HIP(hipEventRecord(b->ev[2], st));
    // ---- the search: one work-queue launch, longest utterance first
    n_main = n - Kh;
    tasks.resize(n_main);
    std::vector<int> order(n_main);
    for (int u = 0; u < n_main; u++) order[u] = Kh + u;
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
      return b->out_off[a + 1] - b->out_off[a] > b->out_off[c + 1] - b->out_off[c];
    });
    for (int k = 0; k < n_main; k++) {
      const int u = order[k];
      const int64_t row = (b->d_ll_override && host_mode && b->load_row.size() == static_cast<size_t>(n)) ? b->load_row[u] : b->out_off[u];
      tasks[k].d_loglikes = ll_base + static_cast<size_t>(row) * b->P;
      tasks[k].ld = b->P; tasks[k].n_frames = static_cast<int32_t>(b->out_off[u + 1] - b->out_off[u]);
      tasks[k].utt = u - Kh; tasks[k].reserved = 0;       // queue-local number (map_rest when the long ones went elsewhere)
      b->task_ll[u] = tasks[k].d_loglikes;
    }
    rc = kamd_decoder_queue_launch(b->dec, tasks.data(), n_main, b->opts.resident_lanes, st);
    if (rc != KAMD_OK) return rc;
    mark("search issued, utterances", n);
  } else {
    // the K longest utterances, longest first; everybody else in input order.  Log-likelihood rows: the long ones first
    // (the forward's output rows are a running sum over its items), then the rest.
    const int K = b->long_lanes;
    std::vector<int> order(n);
    for (int u = 0; u < n; u++) order[u] = u;
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
      return b->out_off[a + 1] - b->out_off[a] > b->out_off[c + 1] - b->out_off[c];
    });
    b->map_long.assign(order.begin(), order.begin() + K);
    std::vector<char> is_long(n, 0);
    for (int k : b->map_long) is_long[k] = 1;
    b->map_rest.clear();
    for (int k = 0; k < n; k++) if (!is_long[k]) b->map_rest.push_back(k);
    int64_t row = 0;
    for (int k : b->map_long) { b->ll_row[k] = row; row += b->out_off[k + 1] - b->out_off[k]; }
    for (int k : b->map_rest) { b->ll_row[k] = row; row += b->out_off[k + 1] - b->out_off[k]; }
    auto forward = [&](const std::vector<int> &items, size_t i0, size_t i1) -> int {
      std::vector<int64_t> in_start(i1 - i0), out_row(i1 - i0 + 1);
      std::vector<int32_t> in_len(i1 - i0);
      for (size_t i = i0; i < i1; i++) {
        const int k = items[i];
        in_start[i - i0] = b->feat_off[k]; in_len[i - i0] = static_cast<int32_t>(b->feat_off[k + 1] - b->feat_off[k]);
        out_row[i - i0] = b->ll_row[k];
      }
      const int kl = items[i1 - 1];
      out_row[i1 - i0] = b->ll_row[kl] + (b->out_off[kl + 1] - b->out_off[kl]);
      const int frc = kamd_nnet_forward_slices_device(b->nnet, b->d_feats, in_start.data(), in_len.data(), b->ld_feat, NULL,
                                                      static_cast<int>(i1 - i0), b->d_ll, out_row.data(), b->P, st);
      if (frc == KAMD_OK) { flops += kamd_nnet_last_flops(b->nnet); passes++; }
      return frc;
    };
    auto make_tasks = [&](const std::vector<int> &items, bool longest_first) {
      std::vector<int> loc(items.size());
      for (size_t i = 0; i < items.size(); i++) loc[i] = static_cast<int>(i);
      if (longest_first)
        std::stable_sort(loc.begin(), loc.end(), [&](int a, int c) {
          return b->out_off[items[a] + 1] - b->out_off[items[a]] > b->out_off[items[c] + 1] - b->out_off[items[c]];
        });
      tasks.resize(items.size());
      for (size_t i = 0; i < items.size(); i++) {
        const int k = items[loc[i]];
        tasks[i].d_loglikes = b->d_ll + static_cast<size_t>(b->ll_row[k]) * b->P;
        tasks[i].ld = b->P; tasks[i].n_frames = static_cast<int32_t>(b->out_off[k + 1] - b->out_off[k]);
        tasks[i].utt = loc[i]; tasks[i].reserved = 0;
        b->task_ll[k] = tasks[i].d_loglikes;
      }
    };
    rc = forward(b->map_long, 0, b->map_long.size());
    if (rc != KAMD_OK) return rc;
    KAMD_HIP(hipEventRecord(b->ev_long, st));
    KAMD_HIP(hipStreamWaitEvent(b->s_long, b->ev_long, 0));
    make_tasks(b->map_long, false);
    rc = kamd_decoder_queue_launch(b->dec_long, tasks.data(), K, K, b->s_long);
    if (rc != KAMD_OK) return rc;
    for (size_t i0 = 0; i0 < b->map_rest.size();) {
      size_t i1 = i0 + 1;
      int64_t fr = b->feat_off[b->map_rest[i0] + 1] - b->feat_off[b->map_rest[i0]];
      while (i1 < b->map_rest.size() && fr + (b->feat_off[b->map_rest[i1] + 1] - b->feat_off[b->map_rest[i1]]) <= b->opts.nnet_pass_frames) {
        fr += b->feat_off[b->map_rest[i1] + 1] - b->feat_off[b->map_rest[i1]];
        i1++;
      }
      rc = forward(b->map_rest, i0, i1);
      if (rc != KAMD_OK) return rc;
      i0 = i1;
    }
    KAMD_HIP(hipEventRecord(b->ev[2], st));
    make_tasks(b->map_rest, true);
    n_main = static_cast<int>(b->map_rest.size());
    rc = kamd_decoder_queue_launch(b->dec, tasks.data(), n_main, b->opts.resident_lanes, st);
    if (rc != KAMD_OK) return rc;
  }
  // ---- collector: hand finished utterances to the host-tail pool while the kernel runs
  int collected = 0, idle_after_end = 0;
  std::vector<int32_t> buf(256);
  double t_first_done = -1, t_last_done = -1;
  while (collected < n) {
    int k = kamd_decoder_queue_poll(b->dec, buf.data(), static_cast<int>(buf.size()));
    if (k == 0 && two_queues) {
      k = kamd_decoder_queue_poll(b->dec_long, buf.data(), static_cast<int>(buf.size()));
      for (int i = 0; i < k; i++) buf[i] |= KAMD_JOB_LONG;
    }
    if (k > 0) {
      const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (t_first_done < 0) t_first_done = now;
      t_last_done = now;
      {
        std::lock_guard<std::mutex> lk(b->mu);
        for (int i = 0; i < k; i++) b->jobs.push_back(buf[i]);
        b->pending += k;
      }
      b->cv_job.notify_all();
      collected += k;
      idle_after_end = 0;
      continue;
    }
    hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess && two_queues) q = hipStreamQuery(b->s_long);
    if (q == hipSuccess) {
      if (++idle_after_end > 2) break;      // the kernel has ended and published nothing more
    } else if (q != hipErrorNotReady) {
      return kamd::SetError(KAMD_ERR_HIP, "decoder work queue failed: %s", hipGetErrorString(q));
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  {
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv_done.wait(lk, [&] { return b->pending == 0; });
  }
  const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  float qms = 0; int32_t lanes = 0;
  rc = kamd_decoder_queue_wait(b->dec, &qms, &lanes);
  if (rc != KAMD_OK) return rc;
  if (two_queues) {
    float qms_long = 0; int32_t lanes_long = 0;
    rc = kamd_decoder_queue_wait(b->dec_long, &qms_long, &lanes_long);
    if (rc != KAMD_OK) return rc;
    lanes += lanes_long;
  }
  if (collected < n) return kamd::SetError(KAMD_ERR_STATE, "work queue ended with %d of %d utterances published", collected, n);
  // ---- second chance: an utterance whose lane ran out of token / link arena (or that found the lattice pool full) is
  // searched again on a lane that owns a larger share of the pools -- a launch for the few of them, the pools split
  // between 1/8 of the lanes first, then 1/64 (kamd_decoder_queue_launch_wide).  The reference has no capacities to run out
  // of; here an unusually dense utterance costs a second search instead of a failure (KAMD_BATCH_RETRY=0 switches it off).
  int n_retried = 0, n_internal = 0;
  // A lane that stopped on one of its internal consistency checks (flag 32: a table entry beyond the cutoff, a state
  // that is no state of the graph, ...) did not run out of anything: the search itself went wrong on that utterance
  // (DESIGN.md section 8.4).  Such an utterance is searched again like one that ran out of arena -- the search is
  // deterministic, the second one gives the lattice the first should have -- but it is COUNTED and REPORTED on its own
  // (kamd_batch_stats.n_internal_events, a line on stderr): a run that hides it is how the event stayed unseen.
  for (int k = 0; k < n; k++) {
    const kamd::UttOut &o = b->out[b->kept[k]];
    if (o.status == KAMD_ERR_CAPACITY && (o.rec.error & 32) != 0) {
      n_internal++;
      fprintf(stderr, "kaldi_amd: WARNING: utterance %d: the search lane stopped on an internal consistency check (flags 0x%x, frame %d of %d)\n",
              b->kept[k], o.rec.error, o.rec.n_frames, static_cast<int>(b->out_off[k + 1] - b->out_off[k]));
    }
  }
  const char *retry_env = getenv("KAMD_BATCH_RETRY");        // (read per run: the tests switch it)
  const bool retry_on = !(retry_env && retry_env[0] == '0');
  const int wide_max_frames = kamd_decoder_max_frames(b->dec);
  for (int round = 0; retry_on && round < 2; round++) {
    std::vector<int> again;
    for (int k = 0; k < n; k++) {
      const kamd::UttOut &o = b->out[b->kept[k]];
      // a frame's level-2 share (1: the wide launch addresses the whole table), token arena (2), link arena (4), lattice pool (64), the internal checks (32; counted above); never an utterance of more
      // frames than the decoder holds (8), nor one longer than the main decoder's frame arrays (it was searched on the
      // long-utterance decoder: its failure stays its own)
      if (o.status == KAMD_ERR_CAPACITY && (o.rec.error & (1 | 2 | 4 | 32 | 64)) != 0 && (o.rec.error & 8) == 0 && b->task_ll[k] &&
          b->out_off[k + 1] - b->out_off[k] <= wide_max_frames)
        again.push_back(k);
    }
    if (again.empty()) break;
    const int max_lanes = kamd_decoder_max_lanes(b->dec);
    const int group = std::max(1, max_lanes / (round == 0 ? 8 : 64));
    for (size_t g0 = 0; g0 < again.size(); g0 += group) {
      const int m = static_cast<int>(std::min(again.size() - g0, static_cast<size_t>(group)));
      std::vector<kamd_queue_task> rt(m);
      b->map_retry.assign(again.begin() + g0, again.begin() + g0 + m);
      for (int i = 0; i < m; i++) {
        const int k = b->map_retry[i];
        rt[i].d_loglikes = b->task_ll[k]; rt[i].ld = b->P;
        rt[i].n_frames = static_cast<int32_t>(b->out_off[k + 1] - b->out_off[k]); rt[i].utt = i; rt[i].reserved = 0;
      }
      rc = kamd_decoder_queue_launch_wide(b->dec, rt.data(), m, st);
      if (rc == KAMD_ERR_ARG) {
        // the launch refused these tasks (the frame-length case is filtered above, so this is unexpected): their first
        // failure stands and the run goes on, but not silently
        fprintf(stderr, "kaldi_amd: second-chance launch of %d utterance(s) refused: %s\n", m, kamd_last_error());
        continue;
      }
      if (rc != KAMD_OK) return rc;
      int got = 0;
      while (got < m) {
        int k = kamd_decoder_queue_poll(b->dec, buf.data(), static_cast<int>(buf.size()));
        if (k > 0) {
          {
            std::lock_guard<std::mutex> lk(b->mu);
            for (int i = 0; i < k; i++) b->jobs.push_back(buf[i] | KAMD_JOB_RETRY);
            b->pending += k;
          }
          b->cv_job.notify_all();
          got += k;
          continue;
        }
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) { if (kamd_decoder_queue_poll(b->dec, buf.data(), 0) == 0 && ++idle_after_end > 4) break; }
        else if (q != hipErrorNotReady) return kamd::SetError(KAMD_ERR_HIP, "decoder work queue (second chance) failed: %s", hipGetErrorString(q));
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_done.wait(lk, [&] { return b->pending == 0; });      // (map_retry is reused by the next group)
      }
      float qms2 = 0; int32_t l2 = 0;
      rc = kamd_decoder_queue_wait(b->dec, &qms2, &l2);
      if (rc != KAMD_OK) return rc;
      if (got < m) return kamd::SetError(KAMD_ERR_STATE, "second-chance queue ended with %d of %d utterances published", got, m);
      qms += qms2;
      n_retried += m;
      idle_after_end = 0;
    }
  }
  const double total_ms_all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  kamd_batch_stats s;
  memset(&s, 0, sizeof(s));
  (void)hipEventElapsedTime(&s.feat_ms, b->ev[0], b->ev[1]);
  (void)hipEventElapsedTime(&s.nnet_ms, b->ev[1], b->ev[2]);
  if (host_mode) {
    if (uploader.joinable()) uploader.join();
    s.feat_ms = 0; s.nnet_ms = 0;
    for (int p = 0; p < passes; p++) {
      float f = 0, m = 0;
      (void)hipEventElapsedTime(&f, b->ev_f0[p], b->ev_f1[p]);
      (void)hipEventElapsedTime(&m, online_iv ? b->ev_n0[p] : b->ev_f1[p], b->ev_n1[p]);
      s.feat_ms += f; s.nnet_ms += m;
    }
    s.upload_ms = static_cast<float>(up.done_ms); s.first_pass_start_ms = static_cast<float>(first_pass_start_ms);
    s.upload_wait_ms = static_cast<float>(upload_wait_ms); s.upload_passes = passes;
  }
  if (online_iv) {
    // the model's own time; the extraction is reported on its own
    if (!host_mode) s.nnet_ms = 0;
    for (int p = 0; p < passes; p++) {
      float v = 0, m = 0;
      (void)hipEventElapsedTime(&v, b->ev_iv0[p], b->ev_iv1[p]);
      s.ivector_ms += v;
      if (!host_mode) { (void)hipEventElapsedTime(&m, b->ev_n0[p], b->ev_n1[p]); s.nnet_ms += m; }
    }
    if (b->last_deferred_solve && b->ev_solve[0]) {
      float v = 0;
      (void)hipEventElapsedTime(&v, b->ev_solve[0], b->ev_solve[1]);
      s.ivector_ms += v;
    }
  }
  s.decode_ms = qms; s.total_ms = static_cast<float>(n_retried ? total_ms_all : total_ms);
  s.host_tail_ms = static_cast<float>(total_ms - t_last_done);
  s.n_retried = n_retried;
  s.n_internal_events = n_internal;
  s.first_result_ms = static_cast<float>(t_first_done);
  s.nnet_flops = flops; s.lanes = lanes; s.nnet_passes = passes;
  s.long_utterances = two_queues ? b->long_lanes : 0;
  double host_sum = 0;
  for (int u = 0; u < b->n_utts; u++) {
    if (b->out[u].status != KAMD_OK) s.n_failed++;
    host_sum += b->out[u].host_ms;
  }
  s.host_thread_ms_sum = host_sum;
  b->last = s;
  if (stats) *stats = s;
  return KAMD_OK;
}

static int UttOk(BatchDecoder *b, int u) {
  if (u < 0 || u >= b->n_utts) return kamd::SetError(KAMD_ERR_ARG, "utterance %d outside the loaded set", u);
  if (!b->out[u].done) return kamd::SetError(KAMD_ERR_STATE, "utterance %d has not been decoded", u);
  return KAMD_OK;
}

int kamd_batch_decoder_get_output(kamd_batch_decoder *h, int utt, int32_t *words, int words_cap, int *words_len, int32_t *alignment,
                                  int ali_cap, int *ali_len, float *graph_cost, float *acoustic_cost, kamd_queue_result *record) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (UttOk(b, utt) != KAMD_OK) return KAMD_ERR_ARG;
  const kamd::UttOut &o = b->out[utt];
  if (record) *record = o.rec;
  if (words_len) *words_len = 0;
  if (ali_len) *ali_len = 0;
  if (graph_cost) *graph_cost = INFINITY;
  if (acoustic_cost) *acoustic_cost = INFINITY;
  if (o.status != KAMD_OK) return kamd::SetError(o.status, "%s", o.message.c_str());
  if (!o.have_path) return kamd::SetError(KAMD_ERR_STATE, "utterance %d: empty lattice", utt);
  if (words_len) { *words_len = static_cast<int>(o.words.size()); if (words) memcpy(words, o.words.data(), sizeof(int32_t) * std::min<size_t>(o.words.size(), words_cap)); }
  if (ali_len) { *ali_len = static_cast<int>(o.ali.size()); if (alignment) memcpy(alignment, o.ali.data(), sizeof(int32_t) * std::min<size_t>(o.ali.size(), ali_cap)); }
  if (graph_cost) *graph_cost = o.graph_cost;
  if (acoustic_cost) *acoustic_cost = o.ac_cost;
  return KAMD_OK;
}

int kamd_batch_decoder_get_raw_lattice(kamd_batch_decoder *h, int utt, int32_t *num_states, int32_t *num_arcs, int32_t *start,
                                       const int32_t **state_frame, const int32_t **state_hclg, const float **state_cost,
                                       const float **state_final, const kamd_lat_arc **arcs) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (UttOk(b, utt) != KAMD_OK) return KAMD_ERR_ARG;
  const kamd::UttOut &o = b->out[utt];
  if (o.status != KAMD_OK) return kamd::SetError(o.status, "%s", o.message.c_str());
  if (!b->opts.keep_raw_lattices) return kamd::SetError(KAMD_ERR_STATE, "raw lattices were not kept (kamd_batch_opts.keep_raw_lattices)");
  *num_states = o.num_states; *num_arcs = o.num_arcs; *start = o.start;
  *state_frame = o.st_frame; *state_hclg = o.st_hclg; *state_cost = o.st_cost; *state_final = o.st_final; *arcs = o.arcs;
  return KAMD_OK;
}

const kamd_compact_lattice *kamd_batch_decoder_get_compact_lattice(kamd_batch_decoder *h, int utt) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (UttOk(b, utt) != KAMD_OK) return NULL;
  if (!b->out[utt].clat) kamd::SetError(KAMD_ERR_STATE, "utterance %d has no determinized lattice", utt);
  return b->out[utt].clat;
}

int kamd_batch_decoder_get_loglikes(kamd_batch_decoder *h, int utt, float *out, int rows_cap, int *rows, int *cols) {
  BatchDecoder *b = reinterpret_cast<BatchDecoder *>(h);
  if (utt < 0 || utt >= b->n_utts) return kamd::SetError(KAMD_ERR_ARG, "bad utterance index");
  const std::vector<int>::const_iterator it = std::find(b->kept.begin(), b->kept.end(), utt);     // (load_host may store the longest utterances first)
  *rows = 0; *cols = b->P;
  if (it == b->kept.end() || *it != utt) return KAMD_OK;                 // too short for one frame: no rows
  const size_t k = static_cast<size_t>(it - b->kept.begin());
  const int r = static_cast<int>(b->out_off[k + 1] - b->out_off[k]);
  *rows = r;
  if (r > rows_cap) return kamd::SetError(KAMD_ERR_ARG, "buffer too small");
  const int64_t row0 = b->ll_row.size() == b->kept.size() ? b->ll_row[k] : b->out_off[k];
  KAMD_HIP(hipMemcpy(out, b->d_ll + static_cast<size_t>(row0) * b->P, static_cast<size_t>(r) * b->P * sizeof(float), hipMemcpyDeviceToHost));
  return KAMD_OK;
}

}  // extern "C"
