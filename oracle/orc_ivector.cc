// TEST INFRASTRUCTURE ONLY (see oracle/orc.py): CPU restatement of Kaldi's online i-vector
// extraction as ivector-extract-online2 runs it, one utterance at a time, fresh adaptation state.
// PARITY UNPINNED: the reference cannot be built here and holds no i-vector extractor, UBM or
// golden i-vectors; every function cites the reference code it follows, and tests/test_oracle_
// ivector.py checks the parts that have closed forms (window statistics, posterior pruning,
// conjugate gradient against a direct solve).
//
//   OnlineCmvn::ComputeStatsForFrame / SmoothOnlineCmvnStats / GetFrame  feat/online-feature.cc:325-440
//   ApplyCmvn (means only)                                                 transform/cmvn.cc:64-91
//   OnlineSpliceFrames::GetFrame, OnlineTransform::GetFrame                feat/online-feature.cc:492-531
//   DiagGmm::LogLikelihoods                                                gmm/diag-gmm.cc:546-562
//   VectorToPosteriorEntry                                                 hmm/posterior.cc:440-508
//   OnlineIvectorFeature::UpdateStatsUntilFrame / UpdateStatsForFrames / GetFrame
//                                                                          online2/online-ivector-feature.cc:206-320
//   IvectorExtractor::ComputeDerivedVars(i)                                ivector/ivector-extractor.cc:208-218
//   OnlineIvectorEstimationStats ctor / AccStats / GetIvector              ivector/ivector-extractor.cc:611-668,732-756,786-795
//   LinearCgd                                                              matrix/optimization.cc:453-557
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <utility>
#include <vector>

#include "../include/kaldi_amd.h"

namespace {

typedef kamd_ivector_desc Desc;

inline size_t Tri(int r, int c) { return r >= c ? static_cast<size_t>(r) * (r + 1) / 2 + c : static_cast<size_t>(c) * (c + 1) / 2 + r; }

// y = A x for a packed symmetric matrix (SpMatrix::AddSpVec)
void SpVec(const std::vector<double> &A, const std::vector<double> &x, std::vector<double> *y) {
  const int n = static_cast<int>(x.size());
  for (int r = 0; r < n; r++) {
    double s = 0;
    for (int c = 0; c < n; c++) s += A[Tri(r, c)] * x[c];
    (*y)[r] = s;
  }
}
double Dot(const std::vector<double> &a, const std::vector<double> &b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
  return s;
}

// LinearCgd<double> with LinearCgdOptions{max_iters, max_error = 0, recompute_residual_factor = 0.01}.
// The fallback to an exact solve when the residual got worse (:547-556) is reported, not taken.
int LinearCgd(int max_iters, const std::vector<double> &A, const std::vector<double> &b, std::vector<double> *x,
              int *got_worse) {
  const int M = static_cast<int>(b.size());
  std::vector<double> r(M), p(M), Ap(M);
  SpVec(A, *x, &Ap);
  for (int i = 0; i < M; i++) { p[i] = b[i] - Ap[i]; r[i] = -p[i]; }
  double r_cur = Dot(r, r);
  const double r_initial = r_cur;
  double r_recompute = r_cur;
  const double max_error_sq = std::numeric_limits<double>::min();
  const double residual_factor = 0.01 * 0.01, inv_residual_factor = 1.0 / residual_factor;
  int k = 0;
  for (; k < M + 5 && k != max_iters; k++) {
    SpVec(A, p, &Ap);
    const double alpha = -Dot(p, r) / Dot(p, Ap);
    for (int i = 0; i < M; i++) (*x)[i] += alpha * p[i];
    for (int i = 0; i < M; i++) r[i] += alpha * Ap[i];
    double r_next = Dot(r, r);
    if (r_next < residual_factor * r_recompute || r_next > inv_residual_factor * r_recompute) {
      SpVec(A, *x, &r);
      for (int i = 0; i < M; i++) r[i] -= b[i];
      r_next = Dot(r, r);
      r_recompute = r_next;
    }
    if (r_next <= max_error_sq) break;
    const double beta = r_next / r_cur;
    for (int i = 0; i < M; i++) p[i] = beta * p[i] - r[i];
    r_cur = r_next;
  }
  if (got_worse) *got_worse = (r_cur > r_initial && r_cur > r_initial + 1.0e-10 * Dot(b, b)) ? 1 : 0;
  return k;
}

struct Extractor {
  Desc d;
  int D;                                   // lda_rows = UBM / extractor feature dim
  std::vector<double> U;                   // [G][Q]  packed M^T Sigma_inv M
  std::vector<double> SigmaInvM;           // [G][D][I]
  explicit Extractor(const Desc &desc) : d(desc), D(desc.lda_rows) {
    const int G = d.num_gauss, I = d.ivector_dim, Q = I * (I + 1) / 2, P = D * (D + 1) / 2;
    U.assign(static_cast<size_t>(G) * Q, 0.0);
    SigmaInvM.assign(static_cast<size_t>(G) * D * I, 0.0);
    for (int g = 0; g < G; g++) {
      const double *M = d.M + static_cast<size_t>(g) * D * I, *S = d.sigma_inv + static_cast<size_t>(g) * P;
      double *SM = &SigmaInvM[static_cast<size_t>(g) * D * I];
      for (int a = 0; a < D; a++)
        for (int j = 0; j < I; j++) {
          double s = 0;
          for (int b = 0; b < D; b++) s += S[Tri(a, b)] * M[static_cast<size_t>(b) * I + j];
          SM[static_cast<size_t>(a) * I + j] = s;
        }
      double *Ug = &U[static_cast<size_t>(g) * Q];
      for (int i = 0; i < I; i++)
        for (int j = 0; j <= i; j++) {
          double s = 0;
          for (int a = 0; a < D; a++) s += M[static_cast<size_t>(a) * I + i] * SM[static_cast<size_t>(a) * I + j];
          Ug[Tri(i, j)] = s;
        }
    }
  }
};

// OnlineCmvn over a whole matrix, frame by frame with the sliding window the reference keeps
void OnlineCmvn(const Desc &d, const float *feats, int T, std::vector<float> *out, const double *speaker_stats = NULL) {
  const int dim = d.feat_dim;
  out->assign(static_cast<size_t>(T) * dim, 0.f);
  std::vector<double> sum(dim, 0.0), sum2(dim, 0.0);
  double count = 0;
  const double *gs = d.global_cmvn_stats;
  const bool nv = d.normalize_variance != 0;         // the second row of the statistics (ComputeStatsForFrame :339-350)
  const int sdim = dim + 1;
  for (int t = 0; t < T; t++) {
    for (int k = 0; k < dim; k++) {
      const double v = static_cast<double>(feats[static_cast<size_t>(t) * dim + k]);
      sum[k] += v;
      if (nv) sum2[k] += v * v;
    }
    count += 1.0;
    const int prev = t - d.cmn_window;
    if (prev >= 0) {
      for (int k = 0; k < dim; k++) {
        const double v = static_cast<double>(feats[static_cast<size_t>(prev) * dim + k]);
        sum[k] -= v;
        if (nv) sum2[k] += -1.0 * (v * v);
      }
      count -= 1.0;
    }
    std::vector<double> st(sum), st2(sum2);
    double cnt = count;
    if (cnt < d.cmn_window && speaker_stats != NULL) {      // SmoothOnlineCmvnStats: speaker stats first
      double from_speaker = d.cmn_window - cnt;
      const double speaker_count = speaker_stats[dim];
      if (from_speaker > d.speaker_frames) from_speaker = d.speaker_frames;
      if (from_speaker > speaker_count) from_speaker = speaker_count;
      if (from_speaker > 0.0) {
        for (int k = 0; k < dim; k++) st[k] += from_speaker / speaker_count * speaker_stats[k];
        if (nv) for (int k = 0; k < dim; k++) st2[k] += from_speaker / speaker_count * speaker_stats[sdim + k];
        cnt += from_speaker / speaker_count * speaker_count;
      }
    }
    if (cnt < d.cmn_window) {                 // ... then the global stats
      double from_global = d.cmn_window - cnt;
      const double gcount = gs[dim];
      if (from_global > d.global_frames) from_global = d.global_frames;
      if (from_global > 0.0) {
        for (int k = 0; k < dim; k++) st[k] += from_global / gcount * gs[k];
        if (nv) for (int k = 0; k < dim; k++) st2[k] += from_global / gcount * gs[sdim + k];
        cnt += from_global / gcount * gcount;
      }
    }
    for (int k = 0; k < dim; k++) {
      float v = feats[static_cast<size_t>(t) * dim + k];
      if (d.normalize_mean && nv) {
        // ApplyCmvn with var_norm (transform/cmvn.cc:92-114): norm is a float matrix; MulColsVec, then AddVecToRows
        const double mean = st[k] / cnt;
        double var = (st2[k] / cnt) - mean * mean;
        if (var < 1.0e-20) var = 1.0e-20;
        const double scale = 1.0 / sqrt(var), offset = -(mean * scale);
        v = v * static_cast<float>(scale);
        v = v + static_cast<float>(offset);
      } else if (d.normalize_mean) {
        const float offset = static_cast<float>(-1.0 / cnt * st[k]);   // Vector<float>::AddVec(double alpha, Vector<double>)
        v = v + offset;
      }
      (*out)[static_cast<size_t>(t) * dim + k] = v;
    }
  }
}

// splice (clamped at both ends: all T frames are "ready" for an OnlineMatrixFeature) + affine / linear LDA
void SpliceLda(const Desc &d, const float *src, int T, std::vector<float> *out) {
  const int dim = d.feat_dim, ns = d.splice_left + 1 + d.splice_right, sd = dim * ns;
  const bool affine = d.lda_cols == sd + 1;
  out->assign(static_cast<size_t>(T) * d.lda_rows, 0.f);
  std::vector<float> sp(sd);
  for (int t = 0; t < T; t++) {
    for (int n = 0; n < ns; n++) {
      int t2 = t - d.splice_left + n;
      if (t2 < 0) t2 = 0;
      if (t2 >= T) t2 = T - 1;
      memcpy(&sp[static_cast<size_t>(n) * dim], src + static_cast<size_t>(t2) * dim, sizeof(float) * dim);
    }
    for (int o = 0; o < d.lda_rows; o++) {
      const float *row = d.lda + static_cast<size_t>(o) * d.lda_cols;
      float acc = affine ? row[sd] : 0.f;
      for (int k = 0; k < sd; k++) acc = acc + row[k] * sp[k];
      (*out)[static_cast<size_t>(t) * d.lda_rows + o] = acc;
    }
  }
}

void UbmLogLikes(const Desc &d, const float *x, std::vector<float> *ll) {
  const int D = d.lda_rows;
  ll->resize(d.num_gauss);
  for (int g = 0; g < d.num_gauss; g++) {
    float acc = d.ubm_gconsts[g];
    const float *mi = d.ubm_means_invvars + static_cast<size_t>(g) * D, *iv = d.ubm_inv_vars + static_cast<size_t>(g) * D;
    for (int k = 0; k < D; k++) acc = acc + mi[k] * x[k];
    for (int k = 0; k < D; k++) acc = acc + (-0.5f * iv[k]) * (x[k] * x[k]);
    (*ll)[g] = acc;
  }
}

// VectorToPosteriorEntry; ties in the sort are broken towards the smaller Gaussian index
float PosteriorEntry(const std::vector<float> &ll, int num_gselect, float min_post, std::vector<std::pair<int, float> > *post) {
  const int G = static_cast<int>(ll.size());
  if (num_gselect > G) num_gselect = G;
  std::vector<std::pair<int, float> > temp;
  float max_like = ll[0];
  for (int g = 1; g < G; g++) max_like = std::max(max_like, ll[g]);
  if (min_post != 0.0f) {
    const float cutoff = max_like + logf(min_post);
    for (int g = 0; g < G; g++)
      if (ll[g] > cutoff) temp.push_back(std::make_pair(g, expf(ll[g] - max_like)));
  }
  if (temp.empty()) {
    temp.resize(G);
    for (int g = 0; g < G; g++) temp[g] = std::make_pair(g, expf(ll[g] - max_like));
  }
  std::sort(temp.begin(), temp.end(), [](const std::pair<int, float> &a, const std::pair<int, float> &b) {
    return a.second > b.second || (a.second == b.second && a.first < b.first);
  });
  const size_t n = std::min<size_t>(temp.size(), num_gselect);
  post->assign(temp.begin(), temp.begin() + n);
  float tot = 0;
  for (size_t i = 0; i < post->size(); i++) tot += (*post)[i].second;
  const float cutoff = min_post * tot;
  while (post->size() > 1 && post->back().second < cutoff) { tot -= post->back().second; post->pop_back(); }
  const float inv = 1.0f / tot;
  for (size_t i = 0; i < post->size(); i++) (*post)[i].second *= inv;
  return max_like + logf(tot);
}

struct Stats {                      // OnlineIvectorEstimationStats
  double prior_offset, max_count, num_frames;
  std::vector<double> quad, lin;
  Stats(int I, double po, double mc) : prior_offset(po), max_count(mc), num_frames(0), quad(static_cast<size_t>(I) * (I + 1) / 2, 0.0), lin(I, 0.0) {
    lin[0] += po;
    for (int i = 0; i < I; i++) quad[Tri(i, i)] += 1.0;
  }
};

// AccStats(extractor, features, gauss_post): per Gaussian, the weighted feature sum first
void AccStats(const Extractor &e, Stats *s, const std::vector<const float *> &rows,
              const std::vector<std::vector<std::pair<int, float> > > &post) {
  const int D = e.D, I = e.d.ivector_dim, Q = I * (I + 1) / 2;
  std::map<int, std::pair<float, std::vector<std::pair<int, float> > > > info;   // gauss -> (tot_weight, [(frame, w)])
  for (size_t t = 0; t < post.size(); t++)
    for (size_t j = 0; j < post[t].size(); j++) {
      auto &gi = info[post[t][j].first];
      gi.first += post[t][j].second;
      gi.second.push_back(std::make_pair(static_cast<int>(t), post[t][j].second));
    }
  double tot_weight = 0;
  std::vector<double> wf(D);
  for (auto it = info.begin(); it != info.end(); ++it) {
    std::fill(wf.begin(), wf.end(), 0.0);
    for (size_t k = 0; k < it->second.second.size(); k++) {
      const float *row = rows[it->second.second[k].first];
      const double w = it->second.second[k].second;
      for (int a = 0; a < D; a++) wf[a] += w * row[a];
    }
    const double *SM = &e.SigmaInvM[static_cast<size_t>(it->first) * D * I];
    for (int j = 0; j < I; j++) {
      double acc = 0;
      for (int a = 0; a < D; a++) acc += SM[static_cast<size_t>(a) * I + j] * wf[a];
      s->lin[j] += acc;
    }
    const double tw = it->second.first;
    const double *Ug = &e.U[static_cast<size_t>(it->first) * Q];
    for (int q = 0; q < Q; q++) s->quad[q] += tw * Ug[q];
    tot_weight += tw;
  }
  if (s->max_count > 0.0) {
    const double old_n = s->num_frames, new_n = s->num_frames + tot_weight;
    const double change = std::max(new_n, s->max_count) / s->max_count - std::max(old_n, s->max_count) / s->max_count;
    if (change != 0.0) {
      s->lin[0] += s->prior_offset * change;
      for (int i = 0; i < I; i++) s->quad[Tri(i, i)] += change;
    }
  }
  s->num_frames += tot_weight;
}

}  // namespace

extern "C" {

// diag outputs may be NULL.  norm_lda / raw_lda: [T x lda_rows]; post_g / post_w: [T x num_gselect]
// (gaussian -1 = empty).  Returns the number of i-vector rows written (ceil(T / period)).
// state_in / state_out (either may be NULL = fresh / not wanted): OnlineIvectorExtractorAdaptationState as
// doubles [2 x (feat_dim+1) speaker CMVN stats | packed quadratic term | linear term | num_frames]
// (GetAdaptationState / SetAdaptationState, online2/online-ivector-feature.cc:400-435; without LimitFrames).
int orc_ivector_extract_online(const kamd_ivector_desc *desc, const float *feats, int T, float *out, int out_rows_cap,
                               float *norm_lda, float *raw_lda, int32_t *post_g, float *post_w, int *cg_got_worse,
                               const double *state_in, double *state_out) {
  const Desc &d = *desc;
  if (T <= 0) return 0;
  const int P = d.ivector_period, n_iv = (T + P - 1) / P, I = d.ivector_dim, D = d.lda_rows;
  if (n_iv > out_rows_cap) return -1;
  if (d.normalize_variance && !d.normalize_mean) return -2;     // OnlineCmvn asserts this combination away (:437)
  Extractor e(d);
  std::vector<float> cm, nl, rl;
  const int sdim = d.feat_dim + 1, Qn = I * (I + 1) / 2;
  const bool have_spk = state_in != NULL && state_in[d.feat_dim] > 0.0;
  OnlineCmvn(d, feats, T, &cm, have_spk ? state_in : NULL);
  SpliceLda(d, cm.data(), T, &nl);
  SpliceLda(d, feats, T, &rl);
  if (norm_lda) memcpy(norm_lda, nl.data(), nl.size() * sizeof(float));
  if (raw_lda) memcpy(raw_lda, rl.data(), rl.size() * sizeof(float));
  Stats st(I, d.prior_offset, d.max_count);
  if (state_in != NULL) {
    const double *q = state_in + 2 * sdim;
    st.quad.assign(q, q + Qn);
    st.lin.assign(q + Qn, q + Qn + I);
    st.num_frames = q[Qn + I];
  }
  std::vector<double> cur(I, 0.0);
  std::vector<float> ll;
  int worse_total = 0;
  std::vector<const float *> rows;
  std::vector<std::vector<std::pair<int, float> > > posts;
  for (int t = 0; t < T; t++) {
    rows.push_back(&rl[static_cast<size_t>(t) * D]);
    UbmLogLikes(d, &nl[static_cast<size_t>(t) * D], &ll);
    std::vector<std::pair<int, float> > post;
    PosteriorEntry(ll, d.num_gselect, d.min_post, &post);         // GetMinPost(1.0) = min_post
    for (size_t j = 0; j < post.size(); j++) post[j].second *= d.posterior_scale * 1.0f;
    if (post_g)
      for (int j = 0; j < d.num_gselect; j++) {
        post_g[static_cast<size_t>(t) * d.num_gselect + j] = j < static_cast<int>(post.size()) ? post[j].first : -1;
        post_w[static_cast<size_t>(t) * d.num_gselect + j] = j < static_cast<int>(post.size()) ? post[j].second : 0.f;
      }
    posts.push_back(post);
    if (t % P == 0) {
      AccStats(e, &st, rows, posts);
      rows.clear(); posts.clear();
      if (st.num_frames > 0.0) {                                   // GetIvector
        if (cur[0] == 0.0) cur[0] = st.prior_offset;
        int worse = 0;
        LinearCgd(d.num_cg_iters, st.quad, st.lin, &cur, &worse);
        worse_total += worse;
      } else {
        std::fill(cur.begin(), cur.end(), 0.0);
        cur[0] = st.prior_offset;
      }
      float *o = out + static_cast<size_t>(t / P) * I;
      for (int j = 0; j < I; j++) o[j] = static_cast<float>(cur[j]);       // ivectors_history_: Vector<BaseFloat>
      o[0] = static_cast<float>(static_cast<double>(o[0]) - d.prior_offset);   // (*feat)(0) -= PriorOffset()
    }
  }
  if (cg_got_worse) *cg_got_worse = worse_total;
  if (state_out != NULL) {
    // OnlineCmvn::GetState(T - 1): the incoming speaker stats plus every frame of this utterance
    for (int k = 0; k < 2 * sdim; k++) state_out[k] = state_in != NULL ? state_in[k] : 0.0;
    for (int t = 0; t < T; t++) {
      for (int k = 0; k < d.feat_dim; k++) {
        const double v = feats[static_cast<size_t>(t) * d.feat_dim + k];
        state_out[k] += v; state_out[sdim + k] += v * v;
      }
      state_out[d.feat_dim] += 1.0;
    }
    double *q = state_out + 2 * sdim;
    std::copy(st.quad.begin(), st.quad.end(), q);
    std::copy(st.lin.begin(), st.lin.end(), q + Qn);
    q[Qn + I] = st.num_frames;
  }
  return n_iv;
}

// OnlineIvectorFeature::GetFrame with use_most_recent_ivector = true, called n_calls times (online2/online-
// ivector-feature.cc:245-320): call c asks for frame upto[c] - 1 (the decodable asks for min(most recent input
// frame, NumFramesReady() - 1), nnet3/decodable-online-looped.cc:174-183): the frames not yet seen enter the
// statistics as ONE batch (UpdateStatsForFrames), then num_cg_iters CG steps from the current estimate; a call
// that brings no new frame returns the current estimate.  out: [n_calls][I], prior offset removed.
// All T frames are given; every call's frames must have their splicing context inside the T frames that
// were available when the reference made the call (the caller's schedule guarantees it), so the features
// can be computed once.
int orc_ivector_extract_streaming(const kamd_ivector_desc *desc, const float *feats, int T, const int32_t *upto, int n_calls,
                                  float *out, const double *state_in, double *state_out) {
  const Desc &d = *desc;
  const int I = d.ivector_dim, D = d.lda_rows, sdim = d.feat_dim + 1, Qn = I * (I + 1) / 2;
  Extractor e(d);
  std::vector<float> cm, nl, rl;
  const bool have_spk = state_in != NULL && state_in[d.feat_dim] > 0.0;
  OnlineCmvn(d, feats, T, &cm, have_spk ? state_in : NULL);
  SpliceLda(d, cm.data(), T, &nl);
  SpliceLda(d, feats, T, &rl);
  Stats st(I, d.prior_offset, d.max_count);
  if (state_in != NULL) {
    const double *q = state_in + 2 * sdim;
    st.quad.assign(q, q + Qn); st.lin.assign(q + Qn, q + Qn + I); st.num_frames = q[Qn + I];
  }
  std::vector<double> cur(I, 0.0);
  cur[0] = d.prior_offset;                       // current_ivector_ at construction (online-ivector-feature.cc:396-398)
  std::vector<float> ll;
  int done = 0;
  for (int c = 0; c < n_calls; c++) {
    if (upto[c] > T) return -1;
    if (upto[c] > done) {
      std::vector<const float *> rows;
      std::vector<std::vector<std::pair<int, float> > > posts;
      for (int t = done; t < upto[c]; t++) {
        rows.push_back(&rl[static_cast<size_t>(t) * D]);
        UbmLogLikes(d, &nl[static_cast<size_t>(t) * D], &ll);
        std::vector<std::pair<int, float> > post;
        PosteriorEntry(ll, d.num_gselect, d.min_post, &post);
        for (size_t j = 0; j < post.size(); j++) post[j].second *= d.posterior_scale * 1.0f;
        posts.push_back(post);
      }
      AccStats(e, &st, rows, posts);
      done = upto[c];
      if (st.num_frames > 0.0) {
        if (cur[0] == 0.0) cur[0] = st.prior_offset;
        LinearCgd(d.num_cg_iters, st.quad, st.lin, &cur, NULL);
      } else {
        std::fill(cur.begin(), cur.end(), 0.0);
        cur[0] = st.prior_offset;
      }
    }
    float *o = out + static_cast<size_t>(c) * I;
    for (int j = 0; j < I; j++) o[j] = static_cast<float>(cur[j]);        // feat->CopyFromVec(current_ivector_)
    o[0] = static_cast<float>(static_cast<double>(o[0]) - d.prior_offset);
  }
  if (state_out != NULL) {
    for (int k = 0; k < 2 * sdim; k++) state_out[k] = state_in != NULL ? state_in[k] : 0.0;
    for (int t = 0; t < T; t++) {
      for (int k = 0; k < d.feat_dim; k++) {
        const double v = feats[static_cast<size_t>(t) * d.feat_dim + k];
        state_out[k] += v; state_out[sdim + k] += v * v;
      }
      state_out[d.feat_dim] += 1.0;
    }
    double *q = state_out + 2 * sdim;
    std::copy(st.quad.begin(), st.quad.end(), q);
    std::copy(st.lin.begin(), st.lin.end(), q + Qn);
    q[Qn + I] = st.num_frames;
  }
  return n_calls;
}

// The same with silence weighting: call c is UpdateStatsUntilFrameWeighted(upto[c] - 1) (online2/online-ivector-
// feature.cc:263-306), whose UpdateStatsForFrames (:191-227) takes the merged (frame, weight) pairs
// [wl_off[c], wl_off[c+1]) -- the caller has popped the delta-weight queue and applied MergePairVectorSumming
// (orc.py DeltaWeightQueue).  Per pair: posteriors pruned with GetMinPost(weight) (:176-188), scaled by
// posterior_scale * weight; a zero weight contributes an empty posterior.
int orc_ivector_extract_streaming_weighted(const kamd_ivector_desc *desc, const float *feats, int T, const int32_t *upto, int n_calls,
                                           const int32_t *wl_off, const int32_t *wl_frame, const float *wl_weight, float *out,
                                           const double *state_in, double *state_out) {
  const Desc &d = *desc;
  const int I = d.ivector_dim, D = d.lda_rows, sdim = d.feat_dim + 1, Qn = I * (I + 1) / 2;
  Extractor e(d);
  std::vector<float> cm, nl, rl;
  const bool have_spk = state_in != NULL && state_in[d.feat_dim] > 0.0;
  OnlineCmvn(d, feats, T, &cm, have_spk ? state_in : NULL);
  SpliceLda(d, cm.data(), T, &nl);
  SpliceLda(d, feats, T, &rl);
  Stats st(I, d.prior_offset, d.max_count);
  if (state_in != NULL) {
    const double *q = state_in + 2 * sdim;
    st.quad.assign(q, q + Qn); st.lin.assign(q + Qn, q + Qn + I); st.num_frames = q[Qn + I];
  }
  std::vector<double> cur(I, 0.0);
  cur[0] = d.prior_offset;
  std::vector<float> ll;
  int done = 0;
  for (int c = 0; c < n_calls; c++) {
    if (upto[c] > T) return -1;
    if (upto[c] > done) {                        // "for (; num_frames_stats_ <= frame; ...)": nothing happens without a new frame
      std::vector<const float *> rows;
      std::vector<std::vector<std::pair<int, float> > > posts;
      for (int k = wl_off[c]; k < wl_off[c + 1]; k++) {
        const int t = wl_frame[k];
        const float weight = wl_weight[k];
        if (t < 0 || t >= upto[c]) return -2;
        rows.push_back(&rl[static_cast<size_t>(t) * D]);
        std::vector<std::pair<int, float> > post;
        if (weight != 0.0f) {
          float min_post = d.min_post;           // GetMinPost
          const float abs_weight = std::fabs(weight);
          min_post /= abs_weight;
          if (min_post > 0.99f) min_post = 0.99f;
          UbmLogLikes(d, &nl[static_cast<size_t>(t) * D], &ll);
          PosteriorEntry(ll, d.num_gselect, min_post, &post);
          for (size_t j = 0; j < post.size(); j++) post[j].second *= d.posterior_scale * weight;
        }
        posts.push_back(post);
      }
      AccStats(e, &st, rows, posts);
      done = upto[c];
      if (st.num_frames > 0.0) {
        if (cur[0] == 0.0) cur[0] = st.prior_offset;
        LinearCgd(d.num_cg_iters, st.quad, st.lin, &cur, NULL);
      } else {
        std::fill(cur.begin(), cur.end(), 0.0);
        cur[0] = st.prior_offset;
      }
    }
    float *o = out + static_cast<size_t>(c) * I;
    for (int j = 0; j < I; j++) o[j] = static_cast<float>(cur[j]);
    o[0] = static_cast<float>(static_cast<double>(o[0]) - d.prior_offset);
  }
  if (state_out != NULL) {
    for (int k = 0; k < 2 * sdim; k++) state_out[k] = state_in != NULL ? state_in[k] : 0.0;
    for (int t = 0; t < T; t++) {
      for (int k = 0; k < d.feat_dim; k++) {
        const double v = feats[static_cast<size_t>(t) * d.feat_dim + k];
        state_out[k] += v; state_out[sdim + k] += v * v;
      }
      state_out[d.feat_dim] += 1.0;
    }
    double *q = state_out + 2 * sdim;
    std::copy(st.quad.begin(), st.quad.end(), q);
    std::copy(st.lin.begin(), st.lin.end(), q + Qn);
    q[Qn + I] = st.num_frames;
  }
  return n_calls;
}

// OnlineIvectorExtractorAdaptationState::LimitFrames (online2/online-ivector-feature.cc:96-117) with
// OnlineIvectorEstimationStats::Scale (ivector/ivector-extractor.cc:671-694), in place
void orc_ivector_state_limit_frames(const kamd_ivector_desc *desc, double *state, float max_remembered_frames) {
  const Desc &d = *desc;
  const int sdim = d.feat_dim + 1, I = d.ivector_dim, Qn = I * (I + 1) / 2;
  const float count = static_cast<float>(state[d.feat_dim]);
  if (count > max_remembered_frames)
    for (int k = 0; k < 2 * sdim; k++) state[k] *= max_remembered_frames / count;
  double *quad = state + 2 * sdim, *lin = quad + Qn, *nf = lin + I;
  const float scaled = max_remembered_frames * d.posterior_scale;
  if (*nf > scaled) {
    const double scale = scaled / *nf, old_n = *nf;
    *nf *= scale;
    for (int q = 0; q < Qn; q++) quad[q] *= scale;
    for (int j = 0; j < I; j++) lin[j] *= scale;
    if (d.max_count == 0.0) {
      lin[0] += d.prior_offset * (1.0 - scale);
      for (int i = 0; i < I; i++) quad[Tri(i, i)] += 1.0 - scale;
    } else {
      const double mc = d.max_count;
      const double old_ps = scale * std::max(old_n, mc) / mc, new_ps = std::max(*nf, mc) / mc;
      lin[0] += d.prior_offset * (new_ps - old_ps);
      for (int i = 0; i < I; i++) quad[Tri(i, i)] += new_ps - old_ps;
    }
  }
}

// pieces, for the closed-form tests
int orc_linear_cgd(int max_iters, int n, const double *A_packed, const double *b, double *x) {
  std::vector<double> A(A_packed, A_packed + static_cast<size_t>(n) * (n + 1) / 2), bv(b, b + n), xv(x, x + n);
  int worse = 0;
  const int k = LinearCgd(max_iters, A, bv, &xv, &worse);
  memcpy(x, xv.data(), sizeof(double) * n);
  return worse ? -k - 1 : k;
}
void orc_online_cmvn(const kamd_ivector_desc *desc, const float *feats, int T, float *out) {
  std::vector<float> o;
  OnlineCmvn(*desc, feats, T, &o);
  memcpy(out, o.data(), o.size() * sizeof(float));
}
float orc_posterior_entry(const float *loglikes, int n, int num_gselect, float min_post, int32_t *gauss, float *post, int *count) {
  std::vector<float> ll(loglikes, loglikes + n);
  std::vector<std::pair<int, float> > p;
  const float r = PosteriorEntry(ll, num_gselect, min_post, &p);
  *count = static_cast<int>(p.size());
  for (size_t i = 0; i < p.size(); i++) { gauss[i] = p[i].first; post[i] = p[i].second; }
  return r;
}

}  // extern "C"
