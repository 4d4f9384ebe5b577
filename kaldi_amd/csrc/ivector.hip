// ivector.hip -- online i-vector extraction on the device (gfx950), what ivector-extract-online2
// computes per utterance (online2/online-ivector-feature.cc:206-320; see include/kaldi_amd.h).
//
// Five kernels per batch of utterances, features already in HBM (written by feat.hip):
//   PrefixKernel   per utterance: running sums of the base features in double -> the sliding-window
//                  statistics of OnlineCmvn for any frame are S[t] - S[t - window]
//   FrontKernel    per 16-frame tile: CMVN (window sums smoothed with the global stats), splice
//                  (clamped at the utterance's ends), LDA; both the normalised and the raw variant
//                  from one staging of the tile in LDS
//   PostKernel     one wavefront per frame: diagonal-UBM log-likelihoods (lane = Gaussian, the
//                  UBM stored transposed so lanes read consecutive floats), top num_gselect by
//                  repeated wave arg-max, VectorToPosteriorEntry's pruning and renormalisation
//   StepStatsKernel one workgroup per i-vector step of every utterance: that step's posterior-weighted
//                  U_g (fp64, 40 KB per selected Gaussian, L2 / Infinity-Cache resident) and
//                  Sigma_inv_M_g^T x, written as per-step increments
//   SolveKernel    one workgroup per utterance, sequential over the i-vector steps: adds the step's
//                  increment (prefetched into registers during the previous step's CG) to the
//                  quadratic / linear terms held in LDS, then num_cg_iters conjugate-gradient steps by
//                  wavefront 0 (rows in lanes, packed symmetric matrix in LDS, wave reductions; no
//                  workgroup barrier inside the CG loop)
// HBM-bound where it is bound at all: 40 KB of U_g per selected Gaussian and frame.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "common.h"
#include "meta_ring.h"

namespace kamd {

constexpr int IV_FT = 16;        // frames per FrontKernel tile
constexpr int IV_MAX_NG = 8;     // num_gselect
constexpr int IV_MAX_DIM = 128;  // ivector_dim (two rows per lane of one wavefront)

struct IvDev {
  int feat_dim, L, R, sd, D, affine;
  const float *ldaT;             // [sd (+1)][D]
  const double *gsum;            // [feat_dim] global sums, then [feat_dim] global sums of squares
  int normalize_variance, sw;    // OnlineCmvn --norm-vars; width of a row of S: feat_dim, or 2 * feat_dim (sums | sums of squares)
  double gcount;
  int cmn_window, speaker_frames, global_frames, normalize_mean;
  int G;
  const float *gconsts, *mivT, *ivT;   // [G], [D][G], [D][G] (inv_vars premultiplied by -0.5)
  int I, Q;
  const double *U, *SM;          // [G][Q], [G][D][I]
  double prior_offset, max_count;
  int period, ng, cg_iters;
  float min_post, log_min_post, post_scale;
};

struct IvUtt {                    // one per utterance (or stream) of a call
  int64_t feat_row;               // its first row in the feature matrix
  int64_t ws_row;                 // its first row in the workspaces (S, LDA outputs, posteriors)
  int64_t out_row;                // its first output row
  int64_t inc_row;                // its first row in the per-step increments
  int T;                          // base frames available (splicing clamps at T - 1)
  int proc_first, proc_end;       // frames whose LDA features and posteriors this call computes
  int stats_first, stats_end;     // mode 1: frames that enter the statistics in this call
  int mode;                       // 0: ivector-extract-online2 stepping (step i = frames (i-1)P+1 .. iP, CG at every step, a row per step)
                                  // 1: streaming GetFrame: steps of P frames from stats_first, CG after the last one only, one row
                                  // 2: streaming GetFrame with silence weighting (UpdateStatsUntilFrameWeighted): the statistics take the
                                  //    (frame, delta weight) entries [wl_off, wl_off + wl_n) of the call's list, P entries per step
  int n_steps;
  int wl_off, wl_n;
  int state_idx;                  // record in the state arrays, or -1
};

struct IvBatch {
  const float *feats; int ld;
  const IvUtt *utt;              // [n]
  double *S;                     // [rows][feat_dim]
  float *norm_lda, *raw_lda;     // [rows][D]
  int32_t *post_g; float *post_w;  // [rows][ng]
  float *out;                    // [rows][I]
  double *dquad, *dlin, *dtotw;  // per step: its own statistics [steps][Q], [steps][I], [steps]
  // adaptation state carried over (NULL: fresh / not wanted), records of state_size doubles:
  // 2 x (feat_dim+1) speaker CMVN stats | packed quadratic term | linear term | num_frames [| current estimate]
  const double *state_in; double *state_out; int state_size;
  int x_off;                     // offset of the current estimate inside a record (streaming), or -1
  // mode 2: frame index within the stream, delta weight, GetMinPost(weight) and its log, per list entry
  const int *wl_frame; const float *wl_weight, *wl_minpost, *wl_logminpost;
  int n_utts, xcd_order;         // StepStatsKernel: the number of utterances; its workgroup -> (utterance, step) order
};

// ---------------------------------------------------------------- running sums
__global__ __launch_bounds__(256) void PrefixKernel(IvDev d, IvBatch b) {
  __shared__ double csum[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const IvUtt ut = b.utt[u];
  const int64_t r0 = ut.feat_row;
  const int T = ut.T;
  int dpw = 1;
  while (dpw < d.feat_dim) dpw <<= 1;          // dims padded to a power of two <= 256
  const int chunks = 256 / dpw, k = tid % dpw, c = tid / dpw;
  const int len = (T + chunks - 1) / chunks, t0 = c * len, t1 = min(T, t0 + len);
  __shared__ double csum2[256];
  double s = 0, s2c = 0;
  if (k < d.feat_dim)
    for (int t = t0; t < t1; t++) { const double v = static_cast<double>(b.feats[(r0 + t) * b.ld + k]); s += v; s2c += v * v; }
  csum[tid] = s; csum2[tid] = s2c;
  __syncthreads();
  double run = 0, run2 = 0;
  for (int c2 = 0; c2 < c; c2++) { run += csum[c2 * dpw + k]; run2 += csum2[c2 * dpw + k]; }
  double sq = 0;
  if (k < d.feat_dim)
    for (int t = t0; t < t1; t++) {
      const double v = static_cast<double>(b.feats[(r0 + t) * b.ld + k]);
      run += v; sq += v * v;
      b.S[(ut.ws_row + t) * d.sw + k] = run;
      if (d.normalize_variance) { run2 += v * v; b.S[(ut.ws_row + t) * d.sw + d.feat_dim + k] = run2; }
    }
  if (b.state_out && ut.mode == 0 && ut.state_idx >= 0) {
    // OnlineCmvn::GetState(T - 1) (feat/online-feature.cc:455-475): incoming speaker stats + all T frames
    __syncthreads();
    csum[tid] = sq;
    __syncthreads();
    if (c == 0 && k < d.feat_dim) {
      double s2 = 0, s1 = 0;
      for (int c2 = 0; c2 < chunks; c2++) s2 += csum[c2 * dpw + k];
      s1 = b.S[(ut.ws_row + T - 1) * d.sw + k];
      const int sd1 = d.feat_dim + 1;
      const double *in = b.state_in ? b.state_in + static_cast<size_t>(ut.state_idx) * b.state_size : NULL;
      double *out = b.state_out + static_cast<size_t>(ut.state_idx) * b.state_size;
      out[k] = (in ? in[k] : 0.0) + s1;
      out[sd1 + k] = (in ? in[sd1 + k] : 0.0) + s2;
      if (k == 0) { out[d.feat_dim] = (in ? in[d.feat_dim] : 0.0) + T; out[sd1 + d.feat_dim] = in ? in[sd1 + d.feat_dim] : 0.0; }
    }
  }
}

// ---------------------------------------------------------------- CMVN + splice + LDA
__global__ __launch_bounds__(256) void FrontKernel(IvDev d, IvBatch b) {
  extern __shared__ float tile[];              // raw[(FT+L+R)][dim], nrm[(FT+L+R)][dim]
  const int u = blockIdx.y, tid = threadIdx.x;
  const IvUtt ut = b.utt[u];
  const int64_t r0 = ut.feat_row;
  const int T = ut.T;
  const int t0 = ut.proc_first + blockIdx.x * IV_FT;
  if (t0 >= ut.proc_end) return;
  const int dim = d.feat_dim, span = IV_FT + d.L + d.R;
  float *raw = tile, *nrm = tile + span * dim;
  for (int i = tid; i < span * dim; i += 256) {
    const int s = i / dim, k = i - s * dim;
    int t2 = t0 - d.L + s;
    t2 = max(0, min(T - 1, t2));
    const float v = b.feats[(r0 + t2) * b.ld + k];
    raw[i] = v;
    float nv = v;
    if (d.normalize_mean) {
      // OnlineCmvn: window [t2 - W + 1, t2], topped up with the global stats (feat/online-feature.cc:325-407)
      const double *S = b.S + ut.ws_row * d.sw;
      const bool nvar = d.normalize_variance != 0;
      double win = S[static_cast<size_t>(t2) * d.sw + k], win2 = nvar ? S[static_cast<size_t>(t2) * d.sw + dim + k] : 0.0;
      double cnt = t2 + 1;
      if (t2 - d.cmn_window >= 0) {
        win -= S[static_cast<size_t>(t2 - d.cmn_window) * d.sw + k];
        if (nvar) win2 -= S[static_cast<size_t>(t2 - d.cmn_window) * d.sw + dim + k];
        cnt = d.cmn_window;
      }
      if (cnt < d.cmn_window && b.state_in != NULL && ut.state_idx >= 0) {   // speaker stats of the carried-over state first
        const double *sp = b.state_in + static_cast<size_t>(ut.state_idx) * b.state_size;
        const double speaker_count = sp[dim];
        double from_speaker = d.cmn_window - cnt;
        if (from_speaker > d.speaker_frames) from_speaker = d.speaker_frames;
        if (from_speaker > speaker_count) from_speaker = speaker_count;
        if (from_speaker > 0.0) {
          win += from_speaker / speaker_count * sp[k];
          if (nvar) win2 += from_speaker / speaker_count * sp[dim + 1 + k];
          cnt += from_speaker / speaker_count * speaker_count;
        }
      }
      if (cnt < d.cmn_window) {
        double from_global = d.cmn_window - cnt;
        if (from_global > d.global_frames) from_global = d.global_frames;
        if (from_global > 0.0) {
          win += from_global / d.gcount * d.gsum[k];
          if (nvar) win2 += from_global / d.gcount * d.gsum[dim + k];
          cnt += from_global / d.gcount * d.gcount;
        }
      }
      if (nvar) {
        // ApplyCmvn with var_norm (transform/cmvn.cc:92-114): a float norm matrix, MulColsVec then AddVecToRows
        const double mean = win / cnt;
        double var = (win2 / cnt) - mean * mean;
        if (var < 1.0e-20) var = 1.0e-20;
        const double scale = 1.0 / sqrt(var), offset = -(mean * scale);
        nv = v * static_cast<float>(scale);
        nv = nv + static_cast<float>(offset);
      } else nv = v + static_cast<float>(-1.0 / cnt * win);
    }
    nrm[i] = nv;
  }
  __syncthreads();
  const int nf = min(IV_FT, ut.proc_end - t0);
  // an item = (variant, output dimension o, four consecutive frames): the LDA coefficient a thread loads serves four frames
  // (the kernel is a chain of dependent multiply-adds per output: 1400 per thread with one frame per item, 560 now); every
  // output's sum runs over k in the same order as before
  constexpr int FQ = 4;
  const int nq = (nf + FQ - 1) / FQ;
  for (int i = tid; i < 2 * nq * d.D; i += 256) {
    const int which = i / (nq * d.D), rem = i - which * nq * d.D;
    const int fq = rem / d.D, o = rem - fq * d.D;
    const float *src = (which ? raw : nrm) + fq * FQ * dim;   // spliced vector of frame f = rows f .. f + L + R of the tile
    const float a0 = d.affine ? d.ldaT[static_cast<size_t>(d.sd) * d.D + o] : 0.f;
    float acc[FQ];
#pragma unroll
    for (int j = 0; j < FQ; j++) acc[j] = a0;
    for (int k = 0; k < d.sd; k++) {
      const float c = d.ldaT[static_cast<size_t>(k) * d.D + o];
#pragma unroll
      for (int j = 0; j < FQ; j++) acc[j] = acc[j] + c * src[min(j, IV_FT - 1 - fq * FQ) * dim + k];     // (rows of the tile only: the clamp repeats the last one)
    }
#pragma unroll
    for (int j = 0; j < FQ; j++)
      if (fq * FQ + j < nf) (which ? b.raw_lda : b.norm_lda)[(ut.ws_row + t0 + fq * FQ + j) * d.D + o] = acc[j];
  }
}

// ---------------------------------------------------------------- UBM posteriors
// F frames per wavefront (batch extraction: 4; streaming / weighted lists: 1): every UBM value a lane loads serves F frames --
// the kernel is bound by those loads (164 KB of transposed means / inverse variances per wavefront, from L2), not by the
// arithmetic.  Per frame the sums run in the same order whatever F is.
template <int F>
__global__ __launch_bounds__(256) void PostKernel(IvDev d, IvBatch b) {
  extern __shared__ float plds[];              // per wave: x[F][D], p[F][G]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const IvUtt ut = b.utt[blockIdx.y];
  // posterior of one frame at weight 1, or (mode 2) of one list entry at its delta weight: min_post becomes
  // GetMinPost(weight), the posteriors are scaled by posterior_scale * weight (online-ivector-feature.cc:176-227)
  int frame[F]; float min_post[F], log_min_post[F], weight[F]; bool on[F];
  bool any = false;
#pragma unroll
  for (int f = 0; f < F; f++) {
    min_post[f] = d.min_post; log_min_post[f] = d.log_min_post; weight[f] = 1.0f;
    const int item = (blockIdx.x * 4 + w) * F + f;
    frame[f] = ut.proc_first + item;
    if (ut.mode == 2) {
      on[f] = item < ut.wl_n;
      if (on[f]) {
        frame[f] = b.wl_frame[ut.wl_off + item]; weight[f] = b.wl_weight[ut.wl_off + item];
        min_post[f] = b.wl_minpost[ut.wl_off + item]; log_min_post[f] = b.wl_logminpost[ut.wl_off + item];
      }
    } else {
      on[f] = frame[f] < ut.proc_end;
    }
    if (!on[f]) frame[f] = ut.mode == 2 ? 0 : ut.proc_first;       // (a valid row to read; nothing is written for it)
    any = any || on[f];
  }
  if (!any) return;
  float *x = plds + w * F * (d.D + d.G), *p = x + F * d.D;
#pragma unroll
  for (int f = 0; f < F; f++)
    for (int k = lane; k < d.D; k += 64) x[f * d.D + k] = b.norm_lda[(ut.ws_row + frame[f]) * d.D + k];
  __builtin_amdgcn_wave_barrier();
  float mx[F];
#pragma unroll
  for (int f = 0; f < F; f++) mx[f] = -INFINITY;
  for (int g = lane; g < d.G; g += 64) {        // DiagGmm::LogLikelihoods: means term, then variance term
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; f++) acc[f] = d.gconsts[g];
    for (int k = 0; k < d.D; k++) {
      const float m = d.mivT[static_cast<size_t>(k) * d.G + g];
#pragma unroll
      for (int f = 0; f < F; f++) acc[f] = acc[f] + m * x[f * d.D + k];
    }
    for (int k = 0; k < d.D; k++) {
      const float iv = d.ivT[static_cast<size_t>(k) * d.G + g];
#pragma unroll
      for (int f = 0; f < F; f++) { const float xv = x[f * d.D + k]; acc[f] = acc[f] + iv * (xv * xv); }
    }
#pragma unroll
    for (int f = 0; f < F; f++) { p[f * d.G + g] = acc[f]; mx[f] = fmaxf(mx[f], acc[f]); }
  }
#pragma unroll
  for (int f = 0; f < F; f++) {
    if (!on[f]) continue;                        // (uniform over the wavefront)
    float *pf = p + f * d.G;
    float mxf = mx[f];
    for (int o = 32; o > 0; o >>= 1) mxf = fmaxf(mxf, __shfl_xor(mxf, o, 64));
    // VectorToPosteriorEntry (hmm/posterior.cc:440-508)
    const float cutoff = mxf + log_min_post[f];
    for (int g = lane; g < d.G; g += 64) {
      const float like = pf[g];
      const bool in = min_post[f] == 0.0f || like > cutoff;
      pf[g] = in ? expf(like - mxf) : -1.0f;      // never all out: the maximum itself passes (log(min_post) < 0), or min_post == 0 takes everything
    }
    float sel_p[IV_MAX_NG]; int sel_g[IV_MAX_NG];
    int n = 0;
    bool more = true;
#pragma unroll
    for (int j = 0; j < IV_MAX_NG; j++) {         // top num_gselect, ties -> smaller index
      sel_p[j] = 0.f; sel_g[j] = -1;
      if (j < d.ng && more) {
        float bp = -1.0f; int bg = 0x7fffffff;
        for (int g = lane; g < d.G; g += 64) { const float v = pf[g]; if (v > bp) { bp = v; bg = g; } }
        for (int o = 32; o > 0; o >>= 1) {
          const float op = __shfl_xor(bp, o, 64); const int og = __shfl_xor(bg, o, 64);
          if (op > bp || (op == bp && og < bg)) { bp = op; bg = og; }
        }
        if (bp < 0.0f) more = false;
        else {
          sel_p[j] = bp; sel_g[j] = bg; n = j + 1;
          if ((bg & 63) == lane) pf[bg] = -1.0f;
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    float tot = 0;
#pragma unroll
    for (int j = 0; j < IV_MAX_NG; j++) if (j < n) tot += sel_p[j];
    const float cut2 = min_post[f] * tot;
#pragma unroll
    for (int j = IV_MAX_NG - 1; j >= 1; j--)      // pop from the back while below min_post of the kept mass
      if (j == n - 1 && sel_p[j] < cut2) { tot -= sel_p[j]; n--; }
    const float inv = 1.0f / tot;
    if (lane < d.ng) {
      float wv = 0.f; int gv = -1;
#pragma unroll
      for (int j = 0; j < IV_MAX_NG; j++)
        if (j == lane && j < n) { wv = sel_p[j] * inv; wv *= d.post_scale * weight[f]; gv = sel_g[j]; }
      const int64_t row = ut.ws_row + frame[f];
      b.post_g[row * d.ng + lane] = gv;
      b.post_w[row * d.ng + lane] = wv;
    }
  }
}

// ---------------------------------------------------------------- statistics + CG
__device__ inline int TriIdx(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
__device__ inline double WaveSum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------- per-step statistics
// AccStats (ivector-extractor.cc:611-668) for the frames of ONE i-vector step, every step of every
// utterance in parallel: the 40 KB rows of U_g and the 32 KB of Sigma_inv_M_g per selected Gaussian
// are the bulk of the traffic (L2 / Infinity-Cache resident: 37 MB for the recipe's extractor) and
// must not sit on the per-utterance sequential chain.  grid (max steps, utterances).
// dynamic LDS of StepStatsKernel: the period's frames, {weight, quadratic-term weight} per (frame, Gaussian) pair, the pairs'
// y_k vectors (doubles), and three ints per pair
static inline size_t StepStatsLdsBytes(const IvDev &d) {
  const size_t pair_cap = static_cast<size_t>(d.period) * d.ng;
  return (static_cast<size_t>(d.period) * d.D + 2 * pair_cap + pair_cap * d.D) * sizeof(double) + 3 * pair_cap * sizeof(int);
}
__global__ __launch_bounds__(256) void StepStatsKernel(IvDev d, IvBatch b) {
  extern __shared__ double ss[];               // xf[period][D], pw[cap], uw[cap], yk[cap][D], then int pg[cap], pt[cap], ug[cap]
  // XCD-aware order (b.xcd_order): the hardware deals consecutive workgroups to the 8 XCDs in turn, each with an L2 of its
  // own (4 MB: ~55 Gaussians' U_g + Sigma_inv_M_g rows of the 512).  Neighbouring steps of an utterance select mostly the
  // same Gaussians, so workgroup w -> (XCD w % 8, its w / 8-th workgroup) takes the w / 8-th item of that XCD's eighth of
  // the (utterance, step) grid: the steps that run side by side on one XCD are consecutive steps of one utterance.
  // Utterances are dealt to the XCDs in turn (u % 8: they are stored longest first, every XCD gets its share of long ones);
  // the grid's y extent is rounded up to a multiple of 8 for this.
  int u = blockIdx.y, i = blockIdx.x;
  if (b.xcd_order) {
    const long long w = static_cast<long long>(blockIdx.y) * gridDim.x + blockIdx.x;
    const long long j = w >> 3;
    u = static_cast<int>(j / gridDim.x) * 8 + static_cast<int>(w & 7);
    i = static_cast<int>(j % gridDim.x);
  }
  if (u >= b.n_utts) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const IvUtt ut = b.utt[u];
  const int64_t r0 = ut.ws_row;
  if (i >= ut.n_steps) return;
  const int I = d.I, Q = d.Q, D = d.D;
  // the step's frames: a contiguous range, or (mode 2) `period` entries of the weighted list
  int f0 = 0, nf;
  const int *wl = NULL;
  if (ut.mode == 0) { f0 = i == 0 ? 0 : (i - 1) * d.period + 1; nf = i * d.period - f0 + 1; }
  else if (ut.mode == 1) { f0 = ut.stats_first + i * d.period; nf = min(ut.stats_end, f0 + d.period) - f0; }
  else { wl = b.wl_frame + ut.wl_off + i * d.period; nf = max(0, min(ut.wl_n - i * d.period, d.period)); }
  const int cap = d.period * d.ng;
  double *xf = ss, *pw = xf + d.period * D;
  int *pg = reinterpret_cast<int *>(pw + 2 * cap + cap * D), *pt = pg + cap;
  __shared__ int s_wcnt[4];
  __shared__ double s_part[4][IV_MAX_DIM];
  for (int k = tid; k < nf * D; k += 256) {
    const int t = k / D, c = k - t * D;
    xf[k] = static_cast<double>(b.raw_lda[(r0 + (wl ? wl[t] : f0 + t)) * D + c]);
  }
  // compact the step's (gaussian, weight, frame) triples, in frame order
  int n_pairs = 0;
  for (int base = 0; base < nf * d.ng; base += 256) {
    const int e = base + tid;
    int g = -1; float wv = 0.f;
    if (e < nf * d.ng) {
      const int t = e / d.ng, sl = e - t * d.ng;
      const int64_t pr = (r0 + (wl ? wl[t] : f0 + t)) * d.ng + sl;
      g = b.post_g[pr]; wv = b.post_w[pr];
    }
    const unsigned long long m = __ballot(g >= 0);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = n_pairs;
    for (int w2 = 0; w2 < wave; w2++) off += s_wcnt[w2];
    if (g >= 0) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1));
      pg[pos] = g; pw[pos] = static_cast<double>(wv); pt[pos] = e / d.ng;
    }
    n_pairs += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  // ---- the pairs of one Gaussian merged (neighbouring frames share most of their selected Gaussians: 40 pairs of a
  // step are ~26 Gaussians on the bench's features): U_g and Sigma_inv_M_g are read once per Gaussian of the step, with
  // the summed weight and the weight-summed feature vector.  uw[k], ug[k], yk[k][D]: Gaussian k of the step.
  double *uw = pw + cap, *yk = uw + cap;
  int *ug = pt + cap;
  __shared__ int s_nu;
  int n_u = 0;
  for (int base = 0; base < n_pairs; base += 256) {
    const int e = base + tid;
    bool lead = false;
    double wsum = 0;
    if (e < n_pairs) {
      const int g = pg[e];
      lead = true;
      for (int e2 = 0; e2 < e; e2++) if (pg[e2] == g) { lead = false; break; }
      if (lead) for (int e2 = e; e2 < n_pairs; e2++) if (pg[e2] == g) wsum += pw[e2];
    }
    const unsigned long long m = __ballot(lead);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = n_u;
    for (int w2 = 0; w2 < wave; w2++) off += s_wcnt[w2];
    if (lead) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1));
      ug[pos] = pg[e]; uw[pos] = wsum;
    }
    n_u += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  for (int k = tid; k < n_u * D; k += 256) {
    const int u = k / D, a = k - u * D, g = ug[u];
    double y = 0;
    for (int e = 0; e < n_pairs; e++) if (pg[e] == g) y += pw[e] * xf[pt[e] * D + a];
    yk[k] = y;
  }
  if (tid == 0) s_nu = n_u;
  __syncthreads();
  n_u = s_nu;
  const int64_t row = ut.inc_row + i;
  if ((Q & 1) == 0) {
    // two columns per thread, 16-byte loads (Q even: every row of U and of the increments starts on a 16-byte boundary): the
    // kernel is bound by the bytes it keeps in flight against the Infinity Cache's latency, and this doubles them per
    // instruction.  Same sums in the same order per element.
    for (int q = 2 * tid; q < Q; q += 512) {
      double a0 = 0, a1 = 0;
#pragma unroll 8
      for (int k = 0; k < n_u; k++) {
        const double2 u2 = *reinterpret_cast<const double2 *>(d.U + static_cast<size_t>(ug[k]) * Q + q);
        a0 += uw[k] * u2.x; a1 += uw[k] * u2.y;
      }
      *reinterpret_cast<double2 *>(b.dquad + row * Q + q) = make_double2(a0, a1);
    }
  } else {
    for (int q = tid; q < Q; q += 256) {
      double acc = 0;
#pragma unroll 8
      for (int k = 0; k < n_u; k++) acc += uw[k] * d.U[static_cast<size_t>(ug[k]) * Q + q];
      b.dquad[row * Q + q] = acc;
    }
  }
  if ((I & 1) == 0) {
    // two rows j per lane (16-byte loads of Sigma_inv_M_g's rows: I even), wavefront w the feature dimensions a = w, w + 4, ...;
    // the four partial sums meet in LDS and are added in wavefront order
    const int jp = 2 * (tid & 63), part = tid >> 6;
    double a0 = 0, a1 = 0;
    if (jp < I)
      for (int k = 0; k < n_u; k++) {
        const double *SM = d.SM + static_cast<size_t>(ug[k]) * D * I;
        const double *yr = yk + k * D;
#pragma unroll 5
        for (int a = part; a < D; a += 4) {
          const double2 m2 = *reinterpret_cast<const double2 *>(SM + static_cast<size_t>(a) * I + jp);
          a0 += m2.x * yr[a]; a1 += m2.y * yr[a];
        }
      }
    if (jp < I) { s_part[part][jp] = a0; s_part[part][jp + 1] = a1; }
    __syncthreads();
    if (tid < I) b.dlin[row * I + tid] = ((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid];
  } else {
    const int j = tid & 127, half = tid >> 7;
    double acc = 0;
    if (j < I)
      for (int k = 0; k < n_u; k++) {
        const double *SM = d.SM + static_cast<size_t>(ug[k]) * D * I;
        const double *yr = yk + k * D;
#pragma unroll 8
        for (int a = half; a < D; a += 2) acc += SM[static_cast<size_t>(a) * I + j] * yr[a];
      }
    if (half == 1 && j < I) s_part[0][j] = acc;
    __syncthreads();
    if (half == 0 && j < I) b.dlin[row * I + j] = acc + s_part[0][j];
  }
  if (tid == 0) {
    double tw = 0;
    for (int e = 0; e < n_pairs; e++) tw += pw[e];
    b.dtotw[row] = tw;
  }
}

// ---------------------------------------------------------------- running statistics + CG
// wave-wide sum of a double with DPP row shifts / broadcasts (GFX9 row_shr, row_bcast15 / 31): six
// dependent steps of two 32-bit moves and one add, against twelve LDS-crossbar permutes for
// __shfl_xor.  Returns the total in every lane (read from lane 63).
template <int kCtrl, int kRowMask>
__device__ inline double DppAdd(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, kCtrl, kRowMask, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, kCtrl, kRowMask, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ inline double WaveSumDpp(double v) {
  v = DppAdd<0x111, 0xf>(v);   // row_shr:1
  v = DppAdd<0x112, 0xf>(v);   // row_shr:2
  v = DppAdd<0x114, 0xf>(v);   // row_shr:4
  v = DppAdd<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of every row holds its row's sum
  v = DppAdd<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = DppAdd<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// LDS-only workgroup barrier (cf. decoder.hip): __syncthreads() would also wait for the prefetch
// loads that are deliberately left in flight across the CG loop
__device__ inline void LdsBar() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// n doubles (n even: 16-byte chunks) from global memory into LDS by DMA, the 256 threads of the workgroup together; returns at
// once -- the data is there after s_waitcnt vmcnt(0) + a barrier.  Inline asm like the decoder's RowDma / the GEMM's operand
// ring: the compiler's own tracking of the builtin would drain vmcnt before every later ds_read.
__device__ __forceinline__ void StageDma(double *lds, const double *src, int n) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)lds));
  const int tid = threadIdx.x, chunks = n / 2;                       // 16-byte chunks
  for (int c0 = 0; c0 < chunks; c0 += 256) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(base + static_cast<unsigned>(c0 + (tid & ~63)) * 16u);
    if (c0 + tid < chunks)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src + 2 * (c0 + tid)) : "memory");
  }
}

// One workgroup per utterance, sequential over its i-vector steps.  The symmetric quadratic term
// lives in REGISTERS: wavefront w owns columns [w*CW, (w+1)*CW), lane l rows l and l+64 of them
// (2*CW doubles per thread).  y = A v is four partial products (one per wavefront, v broadcast
// from LDS) and one LDS exchange; wavefront 0 runs LinearCgd itself (vectors in registers, dot
// products by DPP), wavefronts 1-3 serve its matrix-vector requests.  Step i+1's increment is
// fetched into registers while step i iterates and parked in LDS between the two.
constexpr int IV_CW = IV_MAX_DIM / 4;
__global__ __launch_bounds__(256, 2) void SolveKernel(IvDev d, IvBatch b) {
  extern __shared__ double sl[];               // stage[Q], v[IV_MAX_DIM], part[4][IV_MAX_DIM], lin[IV_MAX_DIM]
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const IvUtt ut = b.utt[u];
  const int I = d.I, Q = d.Q;
  double *stage = sl, *vsh = stage + Q, *part = vsh + IV_MAX_DIM, *lin = part + 4 * IV_MAX_DIM;
  __shared__ double s_num_frames, s_totw;
  __shared__ int s_cmd;                         // 1: another matrix-vector product follows, 0: the step's CG is over
  const int ra = lane, rb = lane + 64;
  const bool ha = ra < I, hb = rb < I;
  const int c0 = wave * IV_CW;
  // OnlineIvectorEstimationStats(ivector_dim, prior_offset, max_count) (ivector-extractor.cc:786-795): quadratic = I
  // ... or the statistics the speaker's previous utterance left (SetAdaptationState, online-ivector-feature.cc:427-435)
  const double *rec_in = (b.state_in && ut.state_idx >= 0) ? b.state_in + static_cast<size_t>(ut.state_idx) * b.state_size : NULL;
  const double *sin = rec_in ? rec_in + 2 * (d.feat_dim + 1) : NULL;
  double Aa[IV_CW], Ab[IV_CW];
#pragma unroll
  for (int k = 0; k < IV_CW; k++) {
    const int c = c0 + k;
    Aa[k] = (c == ra && ha) ? 1.0 : 0.0; Ab[k] = (c == rb && hb) ? 1.0 : 0.0;
    if (sin && c < I) {
      if (ha) Aa[k] = sin[ra >= c ? ra * (ra + 1) / 2 + c : c * (c + 1) / 2 + ra];
      if (hb) Ab[k] = sin[rb >= c ? rb * (rb + 1) / 2 + c : c * (c + 1) / 2 + rb];
    }
  }
  for (int j = tid; j < IV_MAX_DIM; j += 256) { lin[j] = (sin && j < I) ? sin[Q + j] : 0.0; vsh[j] = 0.0; }
  if (tid == 0) s_num_frames = sin ? sin[Q + I] : 0.0;
  __syncthreads();
  if (tid == 0 && !sin) lin[0] = d.prior_offset;
  double xa = 0.0, xb = 0.0;                    // wavefront 0: the current estimate (rows ra, rb)
  if (rec_in && b.x_off >= 0) { if (ha) xa = rec_in[b.x_off + ra]; if (hb) xb = rec_in[b.x_off + rb]; }
  const int n_iv = ut.n_steps;
  const int64_t row0 = ut.inc_row;
  for (int q = tid; q < Q; q += 256) stage[q] = b.dquad[row0 * Q + q];
  double nl = tid < I ? b.dlin[row0 * I + tid] : 0.0;
  double ntw = b.dtotw[row0];
  __syncthreads();

  // partial product of this wavefront's columns with v (in vsh), exchanged through `part`
  auto partial = [&]() {
    double sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < IV_CW; k++) { const double vc = vsh[c0 + k]; sa += Aa[k] * vc; sb += Ab[k] * vc; }
    part[wave * IV_MAX_DIM + ra] = sa; part[wave * IV_MAX_DIM + rb] = sb;
  };

  for (int i = 0; i < n_iv; i++) {
    // ---- this step's increment: packed lower triangle in `stage` -> the register tile
    // (the 64 packed-triangle positions are functions of the lane alone: the compiler would compute them once, outside the
    // loop over the steps, and keep 64 VGPRs alive for good -- the row numbers are made opaque per step instead, a few
    // hundred integer instructions against the step's 15 CG iterations)
    int ra_s = ra, rb_s = rb;
    asm volatile("" : "+v"(ra_s), "+v"(rb_s));
#pragma unroll
    for (int k = 0; k < IV_CW; k++) {
      const int c = c0 + k;
      if (c < I) {
        if (ha) Aa[k] += stage[ra_s >= c ? ra_s * (ra_s + 1) / 2 + c : c * (c + 1) / 2 + ra_s];
        if (hb) Ab[k] += stage[rb_s >= c ? rb_s * (rb_s + 1) / 2 + c : c * (c + 1) / 2 + rb_s];
      }
      // (eight columns' reads at a time: left to itself the scheduler issues all 64 reads first and holds 128 more VGPRs)
      if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < I) lin[tid] += nl;
    if (tid == 0) s_totw = ntw;
    // the next step's increment, in flight during this step's CG: its quadratic part travels HBM -> LDS by DMA
    // (global_load_lds_dwordx4: no registers held -- the 66 VGPRs a register prefetch of it took were what kept the kernel
    // at one workgroup per CU), straight into `stage`, which nobody reads between the barrier below and the end of the step
    const bool more = i + 1 < n_iv;
    const int64_t rn = row0 + i + 1;
    if (more) { nl = tid < I ? b.dlin[rn * I + tid] : 0.0; ntw = b.dtotw[rn]; }
    LdsBar();                                    // (everybody has taken this step's increment out of `stage`)
    if (more) StageDma(stage, b.dquad + rn * Q, Q);
    if (d.max_count > 0.0) {
      const double old_n = s_num_frames, new_n = s_num_frames + s_totw;
      const double change = fmax(new_n, d.max_count) / d.max_count - fmax(old_n, d.max_count) / d.max_count;
      if (change != 0.0) {
#pragma unroll
        for (int k = 0; k < IV_CW; k++) { if (ha && c0 + k == ra) Aa[k] += change; if (hb && c0 + k == rb) Ab[k] += change; }
        if (tid == 0) lin[0] += d.prior_offset * change;
      }
    }
    LdsBar();
    if (tid == 0) s_num_frames += s_totw;
    LdsBar();
    const bool have = s_num_frames > 0.0 && (ut.mode == 0 || i == n_iv - 1);   // streaming: one GetIvector, after the last block
    const bool emit = ut.mode == 0 || i == n_iv - 1;
    // ---- GetIvector (ivector-extractor.cc:732-756): LinearCgd from the previous estimate
    if (wave != 0) {
      if (have)
        for (;;) {                               // serve wavefront 0's products until it says the step is over
          LdsBar();                              // B1: v and the command are published
          if (s_cmd == 0) break;
          partial();
          LdsBar();                              // B2: partials are published
        }
    } else {
      if (have) {
        // y = A v for rows ra, rb; v is given distributed (va, vb)
        auto spvec = [&](double va, double vb, double *ya, double *yb) {
          if (ha) vsh[ra] = va;
          if (hb) vsh[rb] = vb;
          if (lane == 0) s_cmd = 1;
          LdsBar();                              // B1
          partial();
          LdsBar();                              // B2
          *ya = ha ? part[ra] + part[IV_MAX_DIM + ra] + part[2 * IV_MAX_DIM + ra] + part[3 * IV_MAX_DIM + ra] : 0.0;
          *yb = hb ? part[rb] + part[IV_MAX_DIM + rb] + part[2 * IV_MAX_DIM + rb] + part[3 * IV_MAX_DIM + rb] : 0.0;
        };
        if (lane == 0 && xa == 0.0) xa = d.prior_offset;       // "(*ivector)(0) == 0.0": better initial guess
        const double la = ha ? lin[ra] : 0.0, lb = hb ? lin[rb] : 0.0;
        double ya, yb;
        spvec(xa, xb, &ya, &yb);
        double pa = ha ? la - ya : 0.0, pb = hb ? lb - yb : 0.0;
        double rra = -pa, rrb = -pb;
        double r_cur = WaveSumDpp(rra * rra + rrb * rrb);
        double r_recompute = r_cur;
        const double residual_factor = 1.0e-4, inv_residual_factor = 1.0e4, max_error_sq = 2.2250738585072014e-308;
        for (int k = 0; k < I + 5 && k != d.cg_iters; k++) {
          spvec(pa, pb, &ya, &yb);
          const double pr = WaveSumDpp(pa * rra + pb * rrb);
          const double pAp = WaveSumDpp(pa * ya + pb * yb);
          const double alpha = -pr / pAp;
          xa += alpha * pa; xb += alpha * pb;
          rra += alpha * ya; rrb += alpha * yb;
          double r_next = WaveSumDpp(rra * rra + rrb * rrb);
          if (r_next < residual_factor * r_recompute || r_next > inv_residual_factor * r_recompute) {
            spvec(xa, xb, &ya, &yb);
            rra = ha ? ya - la : 0.0; rrb = hb ? yb - lb : 0.0;
            r_next = WaveSumDpp(rra * rra + rrb * rrb);
            r_recompute = r_next;
          }
          if (r_next <= max_error_sq) break;
          const double beta = r_next / r_cur;
          pa = beta * pa - rra; pb = beta * pb - rrb;
          r_cur = r_next;
        }
        if (lane == 0) s_cmd = 0;
        LdsBar();                                // the workers' B1: they leave their loop
      } else if (emit) {
        xa = lane == 0 ? d.prior_offset : 0.0; xb = 0.0;
      }
      float *o = b.out + (ut.out_row + (ut.mode == 0 ? i : 0)) * I;
      if (emit) {
      if (ha) { float v = static_cast<float>(xa); if (ra == 0) v = static_cast<float>(static_cast<double>(v) - d.prior_offset); o[ra] = v; }
      if (hb) o[rb] = static_cast<float>(xb);
      }
    }
    // the next step's increment has landed in `stage` (this wave's DMAs: vmcnt; everybody's: the barrier); an odd Q's last
    // double is no 16-byte chunk and follows here (not the recipe's case: Q = 5050)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (more && (Q & 1) && tid == 0) stage[Q - 1] = b.dquad[rn * Q + Q - 1];
    __syncthreads();
  }
  if (b.state_out && ut.state_idx >= 0) {        // GetAdaptationState: the statistics as of the last i-vector
    double *rec = b.state_out + static_cast<size_t>(ut.state_idx) * b.state_size;
    if (wave == 0 && b.x_off >= 0) { if (ha) rec[b.x_off + ra] = xa; if (hb) rec[b.x_off + rb] = xb; }
    double *so = rec + 2 * (d.feat_dim + 1);
#pragma unroll
    for (int k = 0; k < IV_CW; k++) {
      const int c = c0 + k;
      if (c < I) {
        if (ha && ra >= c) so[ra * (ra + 1) / 2 + c] = Aa[k];
        if (hb && rb >= c) so[rb * (rb + 1) / 2 + c] = Ab[k];
      }
    }
    if (tid < I) so[Q + tid] = lin[tid];
    if (tid == 0) so[Q + I] = s_num_frames;
  }
}

// The row-indexed workspaces: running sums, the two LDA outputs and the posteriors of every feature row.  The silence-
// weighted update (mode 2) re-reads the LDA rows of frames it processed on EARLIER calls, so whoever drives such a
// sequence of calls owns one of these (kamd_ivector_workspace_create) and binds it around its calls; the extractor's own
// one serves the stateless entry points.  Growing keeps the old contents.
struct IvWorkspace {
  double *d_S = NULL; size_t S_cap = 0;
  float *d_nl = NULL, *d_rl = NULL; size_t nl_cap = 0, rl_cap = 0;
  int32_t *d_pg = NULL; float *d_pw = NULL; size_t pg_cap = 0, pw_cap = 0;
  void Free() {
    void *ps[] = {d_S, d_nl, d_rl, d_pg, d_pw};
    for (void *q : ps) if (q) (void)hipFree(q);
    d_S = NULL; d_nl = d_rl = NULL; d_pg = NULL; d_pw = NULL; S_cap = nl_cap = rl_cap = pg_cap = pw_cap = 0;
  }
};

struct IvExtractor {
  IvDev dev;
  kamd_ivector_desc desc;
  // device copies
  float *d_ldaT = NULL, *d_gconsts = NULL, *d_mivT = NULL, *d_ivT = NULL;
  double *d_gsum = NULL, *d_U = NULL, *d_SM = NULL;
  // workspaces
  IvWorkspace own, *ws = &own;          // ws: the bound one (kamd_ivector_extractor_bind_workspace)
  int64_t *d_off = NULL; size_t off_cap = 0;
  double *d_dquad = NULL, *d_dlin = NULL, *d_dtotw = NULL; size_t dq_cap = 0, dl_cap = 0, dt_cap = 0;
  double *d_state_in = NULL, *d_state_out = NULL; size_t si_cap = 0, so_cap = 0;
  MetaRing utt_ring;              // the utterance records of a call (meta_ring.h: no copy engine, no host wait)
  int *d_wlf = NULL; float *d_wlw = NULL; size_t wlf_cap = 0, wlw_cap = 0;   // weighted list: frames | weight, min_post, log(min_post)
  int64_t last_rows = 0;
  int64_t reserved_steps = 0;     // kamd_ivector_online_reserve_steps: rows of the step-statistic buffers a stats / solve sequence shares
};

template <typename T>
static int GrowDev(T **p, size_t *cap, size_t need) {
  if (need <= *cap) return KAMD_OK;
  if (*p) (void)hipFree(*p);
  *p = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(p), need * sizeof(T)));
  *cap = need;
  return KAMD_OK;
}
// the same, keeping what the buffer held (the rows of a stream's earlier frames); the copy is ordered on `st`
template <typename T>
static int GrowDevKeep(T **p, size_t *cap, size_t need, hipStream_t st) {
  if (need <= *cap) return KAMD_OK;
  T *q = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&q), need * sizeof(T)));
  if (*p) {
    KAMD_HIP(hipMemcpyAsync(q, *p, *cap * sizeof(T), hipMemcpyDeviceToDevice, st));
    KAMD_HIP(hipStreamSynchronize(st));
    (void)hipFree(*p);
  }
  *p = q; *cap = need;
  return KAMD_OK;
}
template <typename T>
static T *Upload(const std::vector<T> &v) {
  T *p = DevAlloc<T>(v.size());
  if (p && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); p = NULL; }
  return p;
}

}  // namespace kamd
using kamd::IvExtractor;

extern "C" {

kamd_ivector_extractor *kamd_ivector_extractor_create(const kamd_ivector_desc *dp) {
  if (!kamd::RequireDevice()) return NULL;
  const kamd_ivector_desc &d = *dp;
  const int dim = d.feat_dim, ns = d.splice_left + 1 + d.splice_right, sd = dim * ns, D = d.lda_rows, G = d.num_gauss, I = d.ivector_dim;
  auto bad = [](const char *m) { kamd::SetError(KAMD_ERR_ARG, "i-vector extractor: %s", m); return static_cast<kamd_ivector_extractor *>(NULL); };
  if (dim <= 0 || dim > 256 || d.splice_left < 0 || d.splice_right < 0) return bad("bad feature dim / splicing");
  if (d.lda_cols != sd && d.lda_cols != sd + 1) return bad("LDA columns do not match the spliced features");
  if (D <= 0 || G <= 0 || G > 4096 || I <= 0 || I > kamd::IV_MAX_DIM) return bad("dimensions out of range (ivector_dim <= 128, num_gauss <= 4096)");
  if (d.normalize_variance && !d.normalize_mean) return bad("You cannot normalize the variance but not the mean.");
  if (d.num_gselect <= 0 || d.num_gselect > kamd::IV_MAX_NG) return bad("num_gselect must be 1 .. 8");
  if (d.ivector_period <= 0 || d.min_post < 0 || d.min_post >= 0.5f || d.posterior_scale <= 0 || d.posterior_scale > 1.0f) return bad("bad options");
  if (d.cmn_window <= 0 || d.global_frames > d.cmn_window || d.global_cmvn_stats[dim] <= 0.0) return bad("bad CMVN options / global stats");
  IvExtractor *e = new IvExtractor();
  e->desc = d;
  const int Q = I * (I + 1) / 2, P = D * (D + 1) / 2;
  auto tri = [](int r, int c) { return r >= c ? static_cast<size_t>(r) * (r + 1) / 2 + c : static_cast<size_t>(c) * (c + 1) / 2 + r; };
  // IvectorExtractor::ComputeDerivedVars(i) (ivector-extractor.cc:208-218), in double as there
  std::vector<double> U(static_cast<size_t>(G) * Q), SM(static_cast<size_t>(G) * D * I);
  for (int g = 0; g < G; g++) {
    const double *M = d.M + static_cast<size_t>(g) * D * I, *S = d.sigma_inv + static_cast<size_t>(g) * P;
    double *sm = &SM[static_cast<size_t>(g) * D * I];
    for (int a = 0; a < D; a++)
      for (int j = 0; j < I; j++) {
        double s = 0;
        for (int c = 0; c < D; c++) s += S[tri(a, c)] * M[static_cast<size_t>(c) * I + j];
        sm[static_cast<size_t>(a) * I + j] = s;
      }
    double *Ug = &U[static_cast<size_t>(g) * Q];
    for (int i = 0; i < I; i++)
      for (int j = 0; j <= i; j++) {
        double s = 0;
        for (int a = 0; a < D; a++) s += M[static_cast<size_t>(a) * I + i] * sm[static_cast<size_t>(a) * I + j];
        Ug[tri(i, j)] = s;
      }
  }
  std::vector<float> ldaT(static_cast<size_t>(d.lda_cols) * D), mivT(static_cast<size_t>(D) * G), ivT(static_cast<size_t>(D) * G);
  for (int o = 0; o < D; o++)
    for (int k = 0; k < d.lda_cols; k++) ldaT[static_cast<size_t>(k) * D + o] = d.lda[static_cast<size_t>(o) * d.lda_cols + k];
  for (int g = 0; g < G; g++)
    for (int k = 0; k < D; k++) {
      mivT[static_cast<size_t>(k) * G + g] = d.ubm_means_invvars[static_cast<size_t>(g) * D + k];
      ivT[static_cast<size_t>(k) * G + g] = -0.5f * d.ubm_inv_vars[static_cast<size_t>(g) * D + k];
    }
  std::vector<float> gc(d.ubm_gconsts, d.ubm_gconsts + G);
  std::vector<double> gsum(d.global_cmvn_stats, d.global_cmvn_stats + dim);
  gsum.insert(gsum.end(), d.global_cmvn_stats + dim + 1, d.global_cmvn_stats + 2 * dim + 1);     // row 1: sums of squares
  e->d_ldaT = kamd::Upload(ldaT); e->d_gconsts = kamd::Upload(gc); e->d_mivT = kamd::Upload(mivT); e->d_ivT = kamd::Upload(ivT);
  e->d_gsum = kamd::Upload(gsum); e->d_U = kamd::Upload(U); e->d_SM = kamd::Upload(SM);
  if (!e->d_ldaT || !e->d_gconsts || !e->d_mivT || !e->d_ivT || !e->d_gsum || !e->d_U || !e->d_SM) {
    kamd::SetError(KAMD_ERR_HIP, "i-vector extractor: device allocation failed");
    kamd_ivector_extractor_destroy(reinterpret_cast<kamd_ivector_extractor *>(e));
    return NULL;
  }
  kamd::IvDev &v = e->dev;
  v.feat_dim = dim; v.L = d.splice_left; v.R = d.splice_right; v.sd = sd; v.D = D; v.affine = d.lda_cols == sd + 1;
  v.ldaT = e->d_ldaT; v.gsum = e->d_gsum; v.gcount = d.global_cmvn_stats[dim];
  v.cmn_window = d.cmn_window; v.speaker_frames = d.speaker_frames; v.global_frames = d.global_frames; v.normalize_mean = d.normalize_mean;
  v.normalize_variance = d.normalize_variance ? 1 : 0; v.sw = dim * (v.normalize_variance ? 2 : 1);
  v.G = G; v.gconsts = e->d_gconsts; v.mivT = e->d_mivT; v.ivT = e->d_ivT;
  v.I = I; v.Q = Q; v.U = e->d_U; v.SM = e->d_SM;
  v.prior_offset = d.prior_offset; v.max_count = d.max_count;
  v.period = d.ivector_period; v.ng = d.num_gselect; v.cg_iters = d.num_cg_iters;
  // (the attribute is per kernel and process-wide, not per extractor: it only ever grows -- a second, smaller extractor
  // must not lower the cap under the first one's launches)
  static int solve_lds_max = 0, step_lds_max = 0;
  static std::mutex lds_attr_mu;
  std::lock_guard<std::mutex> lds_attr_lock(lds_attr_mu);
  solve_lds_max = std::max(solve_lds_max, static_cast<int>((static_cast<size_t>(Q) + 6 * kamd::IV_MAX_DIM) * sizeof(double)));
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::SolveKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          solve_lds_max) != hipSuccess) {
    kamd::SetError(KAMD_ERR_HIP, "i-vector extractor: cannot reserve LDS for the solver");
    kamd_ivector_extractor_destroy(reinterpret_cast<kamd_ivector_extractor *>(e));
    return NULL;
  }
  {
    // StepStatsKernel's dynamic LDS grows with period * num_gselect * lda_rows: beyond the default 64 KB the launch needs the
    // attribute, beyond a CU's 160 KB the configuration cannot run at all -- say so here, not at the first extraction
    const size_t lds_step = kamd::StepStatsLdsBytes(v);
    if (lds_step > 150 * 1024) {
      kamd::SetError(KAMD_ERR_ARG, "i-vector extractor: ivector_period %d x num_gselect %d x feature dim %d needs %zu bytes of LDS per step (limit %d)",
                     v.period, v.ng, v.D, lds_step, 150 * 1024);
      kamd_ivector_extractor_destroy(reinterpret_cast<kamd_ivector_extractor *>(e));
      return NULL;
    }
    step_lds_max = std::max(step_lds_max, static_cast<int>(lds_step));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::StepStatsKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            step_lds_max) != hipSuccess) {
      kamd::SetError(KAMD_ERR_HIP, "i-vector extractor: cannot reserve LDS for the statistics step");
      kamd_ivector_extractor_destroy(reinterpret_cast<kamd_ivector_extractor *>(e));
      return NULL;
    }
  }
  v.min_post = d.min_post; v.log_min_post = d.min_post > 0 ? logf(d.min_post) : -INFINITY; v.post_scale = d.posterior_scale;
  return reinterpret_cast<kamd_ivector_extractor *>(e);
}

void kamd_ivector_extractor_destroy(kamd_ivector_extractor *h) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (!e) return;
  e->own.Free();
  void *ps[] = {e->d_ldaT, e->d_gconsts, e->d_mivT, e->d_ivT, e->d_gsum, e->d_U, e->d_SM, e->d_off, e->d_dquad, e->d_dlin, e->d_dtotw, e->d_state_in, e->d_state_out, e->d_wlf, e->d_wlw};
  for (void *p : ps) if (p) (void)hipFree(p);
  delete e;
}

int kamd_ivector_dim(const kamd_ivector_extractor *h) { return reinterpret_cast<const IvExtractor *>(h)->dev.I; }
int kamd_ivector_period(const kamd_ivector_extractor *h) { return reinterpret_cast<const IvExtractor *>(h)->dev.period; }
int kamd_ivector_num_ivectors(const kamd_ivector_extractor *h, int num_frames) {
  const int P = reinterpret_cast<const IvExtractor *>(h)->dev.period;
  return num_frames <= 0 ? 0 : (num_frames + P - 1) / P;
}

kamd_ivector_workspace *kamd_ivector_workspace_create(void) { return reinterpret_cast<kamd_ivector_workspace *>(new kamd::IvWorkspace()); }
void kamd_ivector_workspace_destroy(kamd_ivector_workspace *w) {
  kamd::IvWorkspace *x = reinterpret_cast<kamd::IvWorkspace *>(w);
  if (!x) return;
  x->Free();
  delete x;
}
int kamd_ivector_extractor_bind_workspace(kamd_ivector_extractor *h, kamd_ivector_workspace *w) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  e->ws = w ? reinterpret_cast<kamd::IvWorkspace *>(w) : &e->own;
  return KAMD_OK;
}

int kamd_ivector_state_size(const kamd_ivector_extractor *h) {
  const kamd::IvDev &v = reinterpret_cast<const IvExtractor *>(h)->dev;
  return 2 * (v.feat_dim + 1) + v.Q + v.I + 1;
}

int kamd_ivector_extract_online_device(kamd_ivector_extractor *h, const float *d_feats, const int64_t *h_row_off, int ld_feat,
                                       int n_utts, float *d_out, const int64_t *h_out_row_off, void *stream) {
  return kamd_ivector_extract_online_adapt_device(h, d_feats, h_row_off, ld_feat, n_utts, d_out, h_out_row_off, NULL, NULL, stream);
}

// workspaces for ws_rows feature rows and inc_rows statistic increments; uploads the descriptors; launches
// phase: 0 everything; 1 the statistics only (PrefixKernel .. StepStatsKernel: the step increments stay on the device);
// 2 the solver only, over increments an earlier phase-1 call (or several) left -- inc_rows is then the capacity they were
// written under and must not make the buffers move.
static int RunBatch(IvExtractor *e, const float *d_feats, int ld_feat, const std::vector<kamd::IvUtt> &utts, int64_t ws_rows,
                    int64_t inc_rows, float *d_out, const double *d_state_in, double *d_state_out, int state_size, int x_off,
                    hipStream_t st, const std::vector<int> *wl_frame = NULL, const std::vector<float> *wl_weight = NULL, int phase = 0) {
  const kamd::IvDev &v = e->dev;
  const int n = static_cast<int>(utts.size());
  const size_t n_wl = wl_frame ? wl_frame->size() : 0;
  if (n_wl > 0) {
    // OnlineIvectorFeature::GetMinPost (online-ivector-feature.cc:176-188), in BaseFloat like the reference
    std::vector<float> wx(3 * n_wl);
    for (size_t k = 0; k < n_wl; k++) {
      const float w = (*wl_weight)[k], aw = fabsf(w);
      float mp = v.min_post;
      if (aw == 0.0f) mp = 0.99f;
      else { mp /= aw; if (mp > 0.99f) mp = 0.99f; }
      wx[k] = w; wx[n_wl + k] = mp; wx[2 * n_wl + k] = mp > 0 ? logf(mp) : -INFINITY;
    }
    if (kamd::GrowDev(&e->d_wlf, &e->wlf_cap, n_wl) != KAMD_OK) return KAMD_ERR_HIP;
    if (kamd::GrowDev(&e->d_wlw, &e->wlw_cap, 3 * n_wl) != KAMD_OK) return KAMD_ERR_HIP;
    KAMD_HIP(hipMemcpyAsync(e->d_wlf, wl_frame->data(), n_wl * sizeof(int), hipMemcpyHostToDevice, st));
    KAMD_HIP(hipMemcpyAsync(e->d_wlw, wx.data(), 3 * n_wl * sizeof(float), hipMemcpyHostToDevice, st));
    KAMD_HIP(hipStreamSynchronize(st));
  }
  kamd::IvWorkspace *w = e->ws;
  if (phase == 2 && (static_cast<size_t>(inc_rows) * v.Q > e->dq_cap || static_cast<size_t>(inc_rows) * v.I > e->dl_cap || static_cast<size_t>(inc_rows) > e->dt_cap))
    return kamd::SetError(KAMD_ERR_STATE, "i-vector solve: the step statistics of %lld rows were never reserved", static_cast<long long>(inc_rows));
  if (kamd::GrowDevKeep(&w->d_S, &w->S_cap, static_cast<size_t>(ws_rows) * v.sw, st) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDevKeep(&w->d_nl, &w->nl_cap, static_cast<size_t>(ws_rows) * v.D, st) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDevKeep(&w->d_rl, &w->rl_cap, static_cast<size_t>(ws_rows) * v.D, st) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDevKeep(&w->d_pg, &w->pg_cap, static_cast<size_t>(ws_rows) * v.ng, st) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDevKeep(&w->d_pw, &w->pw_cap, static_cast<size_t>(ws_rows) * v.ng, st) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_dquad, &e->dq_cap, static_cast<size_t>(inc_rows) * v.Q) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_dlin, &e->dl_cap, static_cast<size_t>(inc_rows) * v.I) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_dtotw, &e->dt_cap, static_cast<size_t>(inc_rows)) != KAMD_OK) return KAMD_ERR_HIP;
  void *d_utt = NULL;
  if (e->utt_ring.Acquire(utts.data(), n * sizeof(kamd::IvUtt), &d_utt, st) != KAMD_OK) return KAMD_ERR_HIP;
  struct Releaser { kamd::MetaRing &m; hipStream_t s; ~Releaser() { (void)m.Release(s); } } releaser{e->utt_ring, st};
  int max_T = 0, max_proc = 0, max_steps = 0, max_post = 0;
  for (const kamd::IvUtt &u : utts) {
    max_T = std::max(max_T, u.T); max_proc = std::max(max_proc, u.proc_end - u.proc_first); max_steps = std::max(max_steps, u.n_steps);
    max_post = std::max(max_post, u.mode == 2 ? u.wl_n : u.proc_end - u.proc_first);
  }
  kamd::IvBatch b;
  b.feats = d_feats; b.ld = ld_feat; b.utt = static_cast<const kamd::IvUtt *>(d_utt);
  b.S = w->d_S; b.norm_lda = w->d_nl; b.raw_lda = w->d_rl; b.post_g = w->d_pg; b.post_w = w->d_pw; b.out = d_out;
  b.dquad = e->d_dquad; b.dlin = e->d_dlin; b.dtotw = e->d_dtotw;
  b.n_utts = n; b.xcd_order = 0;
  b.state_in = d_state_in; b.state_out = d_state_out; b.state_size = state_size; b.x_off = x_off;
  b.wl_frame = e->d_wlf; b.wl_weight = e->d_wlw; b.wl_minpost = e->d_wlw + n_wl; b.wl_logminpost = e->d_wlw + 2 * n_wl;
  if (phase != 2) hipLaunchKernelGGL(kamd::PrefixKernel, dim3(n), dim3(256), 0, st, v, b);
  if (phase != 2 && max_proc > 0) {
    const size_t lds_front = static_cast<size_t>(2) * (kamd::IV_FT + v.L + v.R) * v.feat_dim * sizeof(float);
    hipLaunchKernelGGL(kamd::FrontKernel, dim3(kamd::CeilDiv(max_proc, kamd::IV_FT), n), dim3(256), lds_front, st, v, b);
  }
  if (phase != 2 && max_post > 0) {
    // four frames per wavefront for a batch of whole utterances (every UBM load serves four frames); one for streaming updates
    bool whole = true;
    for (const kamd::IvUtt &u : utts) whole = whole && u.mode == 0;
    if (whole && max_post >= 64 && static_cast<size_t>(4) * 4 * (v.D + v.G) * sizeof(float) <= 48 * 1024) {
      const size_t lds_post = static_cast<size_t>(4) * 4 * (v.D + v.G) * sizeof(float);
      hipLaunchKernelGGL(kamd::PostKernel<4>, dim3(kamd::CeilDiv(max_post, 16), n), dim3(256), lds_post, st, v, b);
    } else {
      const size_t lds_post = static_cast<size_t>(4) * (v.D + v.G) * sizeof(float);
      hipLaunchKernelGGL(kamd::PostKernel<1>, dim3(kamd::CeilDiv(max_post, 4), n), dim3(256), lds_post, st, v, b);
    }
  }
  if (max_steps > 0) {
    const size_t lds_step = kamd::StepStatsLdsBytes(v);
    static const bool xcd_off = getenv("KAMD_IV_XCD") != NULL && getenv("KAMD_IV_XCD")[0] == '0';     // A/B switch
    b.n_utts = n; b.xcd_order = xcd_off ? 0 : 1;
    if (phase != 2) hipLaunchKernelGGL(kamd::StepStatsKernel, dim3(max_steps, b.xcd_order ? (n + 7) / 8 * 8 : n), dim3(256), lds_step, st, v, b);
    const size_t lds_solve = (static_cast<size_t>(v.Q) + 6 * kamd::IV_MAX_DIM) * sizeof(double);
    if (phase != 1) hipLaunchKernelGGL(kamd::SolveKernel, dim3(n), dim3(256), lds_solve, st, v, b);
  }
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}

int kamd_ivector_extract_online_adapt_device(kamd_ivector_extractor *h, const float *d_feats, const int64_t *h_row_off, int ld_feat,
                                             int n_utts, float *d_out, const int64_t *h_out_row_off, const double *h_state_in,
                                             double *h_state_out, void *stream) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_utts <= 0) return KAMD_OK;
  const kamd::IvDev &v = e->dev;
  if (ld_feat < v.feat_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_feat %d < feature dim %d", ld_feat, v.feat_dim);
  std::vector<kamd::IvUtt> utts(n_utts);
  for (int u = 0; u < n_utts; u++) {
    const int64_t T = h_row_off[u + 1] - h_row_off[u];
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames", u);
    const int n_iv = static_cast<int>((T + v.period - 1) / v.period);
    if (h_out_row_off[u + 1] - h_out_row_off[u] < n_iv)
      return kamd::SetError(KAMD_ERR_ARG, "utterance %d: output rows too few for %lld frames", u, static_cast<long long>(T));
    kamd::IvUtt &x = utts[u];
    x.feat_row = h_row_off[u]; x.ws_row = h_row_off[u] - h_row_off[0]; x.out_row = h_out_row_off[u]; x.inc_row = h_out_row_off[u] - h_out_row_off[0];
    x.T = static_cast<int>(T); x.proc_first = 0; x.proc_end = x.T; x.stats_first = 0; x.stats_end = x.T; x.mode = 0; x.n_steps = n_iv; x.wl_off = 0; x.wl_n = 0;
    x.state_idx = (h_state_in || h_state_out) ? u : -1;
  }
  const int64_t rows = h_row_off[n_utts] - h_row_off[0], iv_rows = h_out_row_off[n_utts] - h_out_row_off[0];
  const int SS = kamd_ivector_state_size(h);
  const double *d_in = NULL; double *d_so = NULL;
  if (h_state_in) {
    if (kamd::GrowDev(&e->d_state_in, &e->si_cap, static_cast<size_t>(n_utts) * SS) != KAMD_OK) return KAMD_ERR_HIP;
    KAMD_HIP(hipMemcpyAsync(e->d_state_in, h_state_in, static_cast<size_t>(n_utts) * SS * 8, hipMemcpyHostToDevice, st));
    KAMD_HIP(hipStreamSynchronize(st));
    d_in = e->d_state_in;
  }
  if (h_state_out) {
    if (kamd::GrowDev(&e->d_state_out, &e->so_cap, static_cast<size_t>(n_utts) * SS) != KAMD_OK) return KAMD_ERR_HIP;
    d_so = e->d_state_out;
  }
  const int rc = RunBatch(e, d_feats, ld_feat, utts, rows, iv_rows, d_out, d_in, d_so, SS, -1, st);
  if (rc != KAMD_OK) return rc;
  e->last_rows = rows;
  if (h_state_out) {
    KAMD_HIP(hipMemcpyAsync(h_state_out, e->d_state_out, static_cast<size_t>(n_utts) * SS * 8, hipMemcpyDeviceToHost, st));
    KAMD_HIP(hipStreamSynchronize(st));
  }
  return KAMD_OK;
}

// kamd_ivector_extract_online_device in two halves, for a caller that walks a test set pass by pass (kamd_batch_decoder):
// the solver is one sequential chain per utterance, so a launch costs its longest utterance's chain whatever else the GPU
// could do -- four passes paid four chains (31 ms of the headline's 98).  With the step statistics of every pass kept
// (7.2 GB for the 4.84 h set) ONE solver launch pays one.  h_out_row_off: the utterances' rows in the WHOLE set's i-vector
// matrix (the increments use the same numbering); reserve first, then `stats` pass by pass, then `solve` once.
int kamd_ivector_online_reserve_steps(kamd_ivector_extractor *h, int64_t total_iv_rows) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  const kamd::IvDev &v = e->dev;
  if (total_iv_rows <= 0) return kamd::SetError(KAMD_ERR_ARG, "i-vector steps to reserve: %lld", static_cast<long long>(total_iv_rows));
  if (kamd::GrowDev(&e->d_dquad, &e->dq_cap, static_cast<size_t>(total_iv_rows) * v.Q) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_dlin, &e->dl_cap, static_cast<size_t>(total_iv_rows) * v.I) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_dtotw, &e->dt_cap, static_cast<size_t>(total_iv_rows)) != KAMD_OK) return KAMD_ERR_HIP;
  e->reserved_steps = total_iv_rows;
  return KAMD_OK;
}

static int OnlineHalf(kamd_ivector_extractor *h, const float *d_feats, const int64_t *h_row_off, int ld_feat, int n_utts, float *d_out,
                      const int64_t *h_out_row_off, hipStream_t st, int phase) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (n_utts <= 0) return KAMD_OK;
  const kamd::IvDev &v = e->dev;
  if (ld_feat < v.feat_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_feat %d < feature dim %d", ld_feat, v.feat_dim);
  std::vector<kamd::IvUtt> utts(n_utts);
  for (int u = 0; u < n_utts; u++) {
    const int64_t T = h_row_off[u + 1] - h_row_off[u];
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames", u);
    const int n_iv = static_cast<int>((T + v.period - 1) / v.period);
    if (h_out_row_off[u + 1] - h_out_row_off[u] < n_iv || h_out_row_off[u] < 0 || h_out_row_off[u] + n_iv > e->reserved_steps)
      return kamd::SetError(KAMD_ERR_ARG, "utterance %d: i-vector rows [%lld, +%d) outside the %lld reserved (kamd_ivector_online_reserve_steps)", u,
                            static_cast<long long>(h_out_row_off[u]), n_iv, static_cast<long long>(e->reserved_steps));
    kamd::IvUtt &x = utts[u];
    x.feat_row = h_row_off[u]; x.ws_row = h_row_off[u] - h_row_off[0]; x.out_row = h_out_row_off[u]; x.inc_row = h_out_row_off[u];
    x.T = static_cast<int>(T); x.proc_first = 0; x.proc_end = x.T; x.stats_first = 0; x.stats_end = x.T; x.mode = 0; x.n_steps = n_iv; x.wl_off = 0; x.wl_n = 0;
    x.state_idx = -1;
  }
  const int64_t rows = phase == 2 ? 1 : h_row_off[n_utts] - h_row_off[0];
  const int rc = RunBatch(e, d_feats, ld_feat, utts, rows, e->reserved_steps, d_out, NULL, NULL, kamd_ivector_state_size(h), -1, st, NULL, NULL, phase);
  if (rc == KAMD_OK && phase == 1) e->last_rows = rows;
  return rc;
}

int kamd_ivector_online_stats_device(kamd_ivector_extractor *h, const float *d_feats, const int64_t *h_row_off, int ld_feat, int n_utts,
                                     const int64_t *h_out_row_off, void *stream) {
  return OnlineHalf(h, d_feats, h_row_off, ld_feat, n_utts, NULL, h_out_row_off, static_cast<hipStream_t>(stream), 1);
}

int kamd_ivector_online_solve_device(kamd_ivector_extractor *h, const int64_t *h_row_off, int n_utts, float *d_out, const int64_t *h_out_row_off,
                                     void *stream) {
  if (!d_out) return kamd::SetError(KAMD_ERR_ARG, "i-vector solve: no output");
  return OnlineHalf(h, NULL, h_row_off, reinterpret_cast<IvExtractor *>(h)->dev.feat_dim, n_utts, d_out, h_out_row_off, static_cast<hipStream_t>(stream), 2);
}

// Streaming form: OnlineIvectorFeature::GetFrame with use_most_recent_ivector = true
// (online2/online-ivector-feature.cc:206-320): item i is a stream whose base features so far are rows
// [h_feat_row[i], + h_n_base[i]) of d_feats (and of the workspaces: the caller reserves ws_rows_total rows);
// frames [h_n_done[i], h_n_upto[i]) are new: their LDA features and posteriors are computed, they enter
// the statistics as one batch, then num_cg_iters CG steps run from the stream's current estimate.
// d_records: the streams' records of kamd_ivector_stream_record_size() doubles (adaptation state | current
// estimate), device resident, updated in place; h_record[i] says which record item i owns.  d_out row i
// receives the estimate (prior offset removed from dimension 0).
int kamd_ivector_stream_record_size(const kamd_ivector_extractor *h) {
  return kamd_ivector_state_size(h) + reinterpret_cast<const IvExtractor *>(h)->dev.I;
}

// a record for a new utterance: the speaker's adaptation state (NULL: fresh) and no estimate yet
int kamd_ivector_stream_record_init(const kamd_ivector_extractor *h, const double *state, double *record) {
  const kamd::IvDev &v = reinterpret_cast<const IvExtractor *>(h)->dev;
  const int SS = kamd_ivector_state_size(h), sdim = v.feat_dim + 1;
  if (state) memcpy(record, state, sizeof(double) * SS);
  else {
    memset(record, 0, sizeof(double) * SS);
    double *quad = record + 2 * sdim, *lin = quad + v.Q;
    for (int i = 0; i < v.I; i++) quad[static_cast<size_t>(i) * (i + 1) / 2 + i] = 1.0;
    lin[0] = v.prior_offset;
  }
  memset(record + SS, 0, sizeof(double) * v.I);
  return KAMD_OK;
}

int kamd_ivector_stream_update_device(kamd_ivector_extractor *h, const float *d_feats, int ld_feat, int64_t ws_rows_total,
                                      const int64_t *h_feat_row, const int32_t *h_n_base, const int32_t *h_n_done,
                                      const int32_t *h_n_upto, const int32_t *h_record, int n, double *d_records, float *d_out,
                                      void *stream) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return KAMD_OK;
  const kamd::IvDev &v = e->dev;
  if (ld_feat < v.feat_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_feat %d < feature dim %d", ld_feat, v.feat_dim);
  std::vector<kamd::IvUtt> utts(n);
  int64_t inc = 0;
  for (int i = 0; i < n; i++) {
    if (h_n_done[i] < 0 || h_n_upto[i] <= h_n_done[i] || h_n_upto[i] > h_n_base[i] || h_feat_row[i] < 0 ||
        h_feat_row[i] + h_n_base[i] > ws_rows_total)
      return kamd::SetError(KAMD_ERR_ARG, "stream item %d: bad frame ranges", i);
    kamd::IvUtt &x = utts[i];
    x.feat_row = h_feat_row[i]; x.ws_row = h_feat_row[i]; x.out_row = i; x.inc_row = inc;
    x.T = h_n_base[i]; x.proc_first = h_n_done[i]; x.proc_end = h_n_upto[i]; x.stats_first = h_n_done[i]; x.stats_end = h_n_upto[i];
    x.mode = 1; x.n_steps = (h_n_upto[i] - h_n_done[i] + v.period - 1) / v.period; x.state_idx = h_record[i];
    x.wl_off = 0; x.wl_n = 0;
    inc += x.n_steps;
  }
  const int RS = kamd_ivector_stream_record_size(h);
  return RunBatch(e, d_feats, ld_feat, utts, ws_rows_total, inc, d_out, d_records, d_records, RS, RS - v.I, st);
}

// The same tick with silence weighting (UpdateStatsUntilFrameWeighted + UpdateStatsForFrames, online2/online-ivector-
// feature.cc:191-227, 263-306): frames [h_n_done[i], h_n_upto[i]) are new (their LDA features are computed and stay
// in the workspace rows for later re-weighting); the statistics take item i's entries [h_wl_off[i], h_wl_off[i+1]) of
// (h_wl_frame, h_wl_weight): frames < h_n_upto[i] in increasing order, no duplicates, no zero weights (the caller has
// applied MergePairVectorSumming), each with its own min_post = GetMinPost(weight) and its posteriors scaled by
// posterior_scale * weight (negative when a frame is re-classified as silence).  An empty list still runs GetIvector.
int kamd_ivector_stream_update_weighted_device(kamd_ivector_extractor *h, const float *d_feats, int ld_feat, int64_t ws_rows_total,
                                               const int64_t *h_feat_row, const int32_t *h_n_base, const int32_t *h_n_done,
                                               const int32_t *h_n_upto, const int32_t *h_record, const int32_t *h_wl_off,
                                               const int32_t *h_wl_frame, const float *h_wl_weight, int n, double *d_records,
                                               float *d_out, void *stream) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0) return KAMD_OK;
  const kamd::IvDev &v = e->dev;
  if (ld_feat < v.feat_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_feat %d < feature dim %d", ld_feat, v.feat_dim);
  std::vector<kamd::IvUtt> utts(n);
  std::vector<int> wlf(h_wl_frame, h_wl_frame + h_wl_off[n]);
  std::vector<float> wlw(h_wl_weight, h_wl_weight + h_wl_off[n]);
  int64_t inc = 0;
  for (int i = 0; i < n; i++) {
    if (h_n_done[i] < 0 || h_n_upto[i] < h_n_done[i] || h_n_upto[i] > h_n_base[i] || h_feat_row[i] < 0 ||
        h_feat_row[i] + h_n_base[i] > ws_rows_total || h_wl_off[i + 1] < h_wl_off[i])
      return kamd::SetError(KAMD_ERR_ARG, "stream item %d: bad frame ranges", i);
    for (int k = h_wl_off[i]; k < h_wl_off[i + 1]; k++)
      if (wlf[k] < 0 || wlf[k] >= h_n_upto[i] || (k > h_wl_off[i] && wlf[k] <= wlf[k - 1]) || wlw[k] == 0.0f)
        return kamd::SetError(KAMD_ERR_ARG, "stream item %d: weighted frames must be increasing, below %d and of non-zero weight", i, h_n_upto[i]);
    kamd::IvUtt &x = utts[i];
    x.feat_row = h_feat_row[i]; x.ws_row = h_feat_row[i]; x.out_row = i; x.inc_row = inc;
    x.T = h_n_base[i]; x.proc_first = h_n_done[i]; x.proc_end = h_n_upto[i]; x.stats_first = 0; x.stats_end = 0;
    x.mode = 2; x.wl_off = h_wl_off[i]; x.wl_n = h_wl_off[i + 1] - h_wl_off[i];
    x.n_steps = std::max(1, (x.wl_n + v.period - 1) / v.period); x.state_idx = h_record[i];
    inc += x.n_steps;
  }
  const int RS = kamd_ivector_stream_record_size(h);
  return RunBatch(e, d_feats, ld_feat, utts, ws_rows_total, inc, d_out, d_records, d_records, RS, RS - v.I, st, &wlf, &wlw);
}

// OnlineIvectorExtractorAdaptationState::LimitFrames (online2/online-ivector-feature.cc:96-117) with
// OnlineIvectorEstimationStats::Scale (ivector/ivector-extractor.cc:671-694): host arithmetic on one state
int kamd_ivector_state_limit_frames(const kamd_ivector_extractor *h, double *state, float max_remembered_frames) {
  const IvExtractor *e = reinterpret_cast<const IvExtractor *>(h);
  const kamd::IvDev &v = e->dev;
  if (max_remembered_frames < 0) return kamd::SetError(KAMD_ERR_ARG, "max_remembered_frames < 0");
  const int sdim = v.feat_dim + 1, I = v.I, Q = v.Q;
  const float count = static_cast<float>(state[v.feat_dim]);
  if (count > max_remembered_frames)
    for (int k = 0; k < 2 * sdim; k++) state[k] *= max_remembered_frames / count;
  double *quad = state + 2 * sdim, *lin = quad + Q, *nf = lin + I;
  const float scaled = max_remembered_frames * v.post_scale;
  if (*nf > scaled) {
    const double scale = scaled / *nf, old_n = *nf;
    *nf *= scale;
    for (int q = 0; q < Q; q++) quad[q] *= scale;
    for (int j = 0; j < I; j++) lin[j] *= scale;
    double add_lin, add_diag;
    if (v.max_count == 0.0) { add_lin = v.prior_offset * (1.0 - scale); add_diag = 1.0 - scale; }
    else {
      const double mc = v.max_count;
      const double old_ps = scale * std::max(old_n, mc) / mc, new_ps = std::max(*nf, mc) / mc;
      add_lin = v.prior_offset * (new_ps - old_ps); add_diag = new_ps - old_ps;
    }
    lin[0] += add_lin;
    for (int i = 0; i < I; i++) quad[static_cast<size_t>(i) * (i + 1) / 2 + i] += add_diag;
  }
  return KAMD_OK;
}

int kamd_ivector_extract_online(kamd_ivector_extractor *h, const float *feats, int num_frames, float *out, int out_rows_cap) {
  return kamd_ivector_extract_online_adapt(h, feats, num_frames, out, out_rows_cap, NULL, NULL);
}

int kamd_ivector_extract_online_adapt(kamd_ivector_extractor *h, const float *feats, int num_frames, float *out, int out_rows_cap,
                                      const double *state_in, double *state_out) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (num_frames <= 0) return 0;
  const int n = kamd_ivector_num_ivectors(h, num_frames), I = e->dev.I, dim = e->dev.feat_dim;
  if (n > out_rows_cap) return kamd::SetError(KAMD_ERR_ARG, "output buffer too small");
  float *d_f = NULL, *d_o = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_f), static_cast<size_t>(num_frames) * dim * sizeof(float)));
  if (hipMalloc(reinterpret_cast<void **>(&d_o), static_cast<size_t>(n) * I * sizeof(float)) != hipSuccess) {
    (void)hipFree(d_f);
    return kamd::SetError(KAMD_ERR_HIP, "allocation failed");
  }
  int rc = KAMD_OK;
  if (hipMemcpy(d_f, feats, static_cast<size_t>(num_frames) * dim * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "upload failed");
  const int64_t ro[2] = {0, num_frames}, oo[2] = {0, n};
  if (rc == KAMD_OK) rc = kamd_ivector_extract_online_adapt_device(h, d_f, ro, dim, 1, d_o, oo, state_in, state_out, NULL);
  if (rc == KAMD_OK && hipMemcpy(out, d_o, static_cast<size_t>(n) * I * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "i-vector extraction failed: %s", hipGetErrorString(hipGetLastError()));
  (void)hipFree(d_f); (void)hipFree(d_o);
  return rc == KAMD_OK ? n : rc;
}

int kamd_ivector_last_posteriors(kamd_ivector_extractor *h, int32_t *gauss, float *weight, int64_t frames_cap) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (frames_cap < e->last_rows) return kamd::SetError(KAMD_ERR_ARG, "buffer too small for %lld frames", static_cast<long long>(e->last_rows));
  KAMD_HIP(hipMemcpy(gauss, e->ws->d_pg, static_cast<size_t>(e->last_rows) * e->dev.ng * 4, hipMemcpyDeviceToHost));
  KAMD_HIP(hipMemcpy(weight, e->ws->d_pw, static_cast<size_t>(e->last_rows) * e->dev.ng * 4, hipMemcpyDeviceToHost));
  return KAMD_OK;
}

}  // extern "C"
