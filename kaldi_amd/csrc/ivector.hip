// ivector.hip -- online i-vector extraction on the device (gfx950), what ivector-extract-online2
// computes per utterance (online2/online-ivector-feature.cc:206-320; see include/kaldi_amd.h).
//
// Four kernels per batch of utterances, features already in HBM (written by feat.hip):
//   PrefixKernel   per utterance: running sums of the base features in double -> the sliding-window
//                  statistics of OnlineCmvn for any frame are S[t] - S[t - window]
//   FrontKernel    per 16-frame tile: CMVN (window sums smoothed with the global stats), splice
//                  (clamped at the utterance's ends), LDA; both the normalised and the raw variant
//                  from one staging of the tile in LDS
//   PostKernel     one wavefront per frame: diagonal-UBM log-likelihoods (lane = Gaussian, the
//                  UBM stored transposed so lanes read consecutive floats), top num_gselect by
//                  repeated wave arg-max, VectorToPosteriorEntry's pruning and renormalisation
//   SolveKernel    one workgroup per utterance, sequential over the i-vector steps: the step's
//                  posterior-weighted U_g (fp64, streamed from L2) and Sigma_inv_M_g^T x into the
//                  quadratic / linear terms held in LDS, then num_cg_iters conjugate-gradient steps by
//                  wavefront 0 (rows in lanes, packed symmetric matrix in LDS, wave reductions; no
//                  workgroup barrier inside the CG loop)
// HBM-bound where it is bound at all: 40 KB of U_g per selected Gaussian and frame.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

namespace kamd {

constexpr int IV_FT = 16;        // frames per FrontKernel tile
constexpr int IV_MAX_NG = 8;     // num_gselect
constexpr int IV_MAX_DIM = 128;  // ivector_dim (two rows per lane of one wavefront)

struct IvDev {
  int feat_dim, L, R, sd, D, affine;
  const float *ldaT;             // [sd (+1)][D]
  const double *gsum;            // [feat_dim] global sums
  double gcount;
  int cmn_window, global_frames, normalize_mean;
  int G;
  const float *gconsts, *mivT, *ivT;   // [G], [D][G], [D][G] (inv_vars premultiplied by -0.5)
  int I, Q;
  const double *U, *SM;          // [G][Q], [G][D][I]
  double prior_offset, max_count;
  int period, ng, cg_iters;
  float min_post, log_min_post, post_scale;
};

struct IvBatch {
  const float *feats; int ld;
  const int64_t *row_off;        // [n + 1] feature rows
  const int64_t *out_off;        // [n + 1] i-vector rows
  double *S;                     // [rows][feat_dim]
  float *norm_lda, *raw_lda;     // [rows][D]
  int32_t *post_g; float *post_w;  // [rows][ng]
  float *out;                    // [iv rows][I]
  int64_t row_base;              // row_off[0]: workspaces are indexed relative to it
};

// ---------------------------------------------------------------- running sums
__global__ __launch_bounds__(256) void PrefixKernel(IvDev d, IvBatch b) {
  __shared__ double csum[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t r0 = b.row_off[u];
  const int T = static_cast<int>(b.row_off[u + 1] - r0);
  int dpw = 1;
  while (dpw < d.feat_dim) dpw <<= 1;          // dims padded to a power of two <= 256
  const int chunks = 256 / dpw, k = tid % dpw, c = tid / dpw;
  const int len = (T + chunks - 1) / chunks, t0 = c * len, t1 = min(T, t0 + len);
  double s = 0;
  if (k < d.feat_dim)
    for (int t = t0; t < t1; t++) s += static_cast<double>(b.feats[(r0 + t) * b.ld + k]);
  csum[tid] = s;
  __syncthreads();
  double run = 0;
  for (int c2 = 0; c2 < c; c2++) run += csum[c2 * dpw + k];
  if (k < d.feat_dim)
    for (int t = t0; t < t1; t++) {
      run += static_cast<double>(b.feats[(r0 + t) * b.ld + k]);
      b.S[(r0 - b.row_base + t) * d.feat_dim + k] = run;
    }
}

// ---------------------------------------------------------------- CMVN + splice + LDA
__global__ __launch_bounds__(256) void FrontKernel(IvDev d, IvBatch b) {
  extern __shared__ float tile[];              // raw[(FT+L+R)][dim], nrm[(FT+L+R)][dim]
  const int u = blockIdx.y, tid = threadIdx.x;
  const int64_t r0 = b.row_off[u];
  const int T = static_cast<int>(b.row_off[u + 1] - r0);
  const int t0 = blockIdx.x * IV_FT;
  if (t0 >= T) return;
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
      const double *S = b.S + (r0 - b.row_base) * dim;
      double win = S[static_cast<size_t>(t2) * dim + k];
      double cnt = t2 + 1;
      if (t2 - d.cmn_window >= 0) { win -= S[static_cast<size_t>(t2 - d.cmn_window) * dim + k]; cnt = d.cmn_window; }
      if (cnt < d.cmn_window) {
        double from_global = d.cmn_window - cnt;
        if (from_global > d.global_frames) from_global = d.global_frames;
        if (from_global > 0.0) { win += from_global / d.gcount * d.gsum[k]; cnt += from_global / d.gcount * d.gcount; }
      }
      nv = v + static_cast<float>(-1.0 / cnt * win);
    }
    nrm[i] = nv;
  }
  __syncthreads();
  const int nf = min(IV_FT, T - t0);
  for (int i = tid; i < 2 * nf * d.D; i += 256) {
    const int which = i / (nf * d.D), rem = i - which * nf * d.D;
    const int f = rem / d.D, o = rem - f * d.D;
    const float *src = (which ? raw : nrm) + f * dim;        // spliced vector = rows f .. f + L + R of the tile
    float acc = d.affine ? d.ldaT[static_cast<size_t>(d.sd) * d.D + o] : 0.f;
    for (int k = 0; k < d.sd; k++) acc = acc + d.ldaT[static_cast<size_t>(k) * d.D + o] * src[k];
    (which ? b.raw_lda : b.norm_lda)[(r0 - b.row_base + t0 + f) * d.D + o] = acc;
  }
}

// ---------------------------------------------------------------- UBM posteriors
__global__ __launch_bounds__(256) void PostKernel(IvDev d, IvBatch b, int64_t rows) {
  extern __shared__ float plds[];              // per wave: x[D], p[G]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + w;
  if (row >= rows) return;
  float *x = plds + w * (d.D + d.G), *p = x + d.D;
  for (int k = lane; k < d.D; k += 64) x[k] = b.norm_lda[row * d.D + k];
  __builtin_amdgcn_wave_barrier();
  float mx = -INFINITY;
  for (int g = lane; g < d.G; g += 64) {        // DiagGmm::LogLikelihoods: means term, then variance term
    float acc = d.gconsts[g];
    for (int k = 0; k < d.D; k++) acc = acc + d.mivT[static_cast<size_t>(k) * d.G + g] * x[k];
    for (int k = 0; k < d.D; k++) acc = acc + d.ivT[static_cast<size_t>(k) * d.G + g] * (x[k] * x[k]);
    p[g] = acc;
    mx = fmaxf(mx, acc);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  // VectorToPosteriorEntry (hmm/posterior.cc:440-508)
  const float cutoff = mx + d.log_min_post;
  int n_cand = 0;
  for (int g = lane; g < d.G; g += 64) {
    const float like = p[g];
    const bool in = d.min_post == 0.0f || like > cutoff;
    p[g] = in ? expf(like - mx) : -1.0f;
    n_cand += in;
  }
  for (int o = 32; o > 0; o >>= 1) n_cand += __shfl_xor(n_cand, o, 64);
  (void)n_cand;   // never 0: the maximum itself passes (log(min_post) < 0), or min_post == 0 takes everything
  float sel_p[IV_MAX_NG]; int sel_g[IV_MAX_NG];
  int n = 0;
  bool more = true;
#pragma unroll
  for (int j = 0; j < IV_MAX_NG; j++) {         // top num_gselect, ties -> smaller index
    sel_p[j] = 0.f; sel_g[j] = -1;
    if (j < d.ng && more) {
      float bp = -1.0f; int bg = 0x7fffffff;
      for (int g = lane; g < d.G; g += 64) { const float v = p[g]; if (v > bp) { bp = v; bg = g; } }
      for (int o = 32; o > 0; o >>= 1) {
        const float op = __shfl_xor(bp, o, 64); const int og = __shfl_xor(bg, o, 64);
        if (op > bp || (op == bp && og < bg)) { bp = op; bg = og; }
      }
      if (bp < 0.0f) more = false;
      else {
        sel_p[j] = bp; sel_g[j] = bg; n = j + 1;
        if ((bg & 63) == lane) p[bg] = -1.0f;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  float tot = 0;
#pragma unroll
  for (int j = 0; j < IV_MAX_NG; j++) if (j < n) tot += sel_p[j];
  const float cut2 = d.min_post * tot;
#pragma unroll
  for (int j = IV_MAX_NG - 1; j >= 1; j--)      // pop from the back while below min_post of the kept mass
    if (j == n - 1 && sel_p[j] < cut2) { tot -= sel_p[j]; n--; }
  const float inv = 1.0f / tot;
  if (lane < d.ng) {
    float wv = 0.f; int gv = -1;
#pragma unroll
    for (int j = 0; j < IV_MAX_NG; j++)
      if (j == lane && j < n) { wv = sel_p[j] * inv; wv *= d.post_scale * 1.0f; gv = sel_g[j]; }
    b.post_g[row * d.ng + lane] = gv;
    b.post_w[row * d.ng + lane] = wv;
  }
}

// ---------------------------------------------------------------- statistics + CG
__device__ inline int TriIdx(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
__device__ inline double WaveSum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void SolveKernel(IvDev d, IvBatch b) {
  extern __shared__ double sl[];               // quad[Q], lin[I], x[I], r[I], p[I], Ap[I], xf[D]
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = b.row_off[u] - b.row_base;
  const int T = static_cast<int>(b.row_off[u + 1] - b.row_off[u]);
  const int I = d.I, Q = d.Q, D = d.D;
  double *quad = sl, *lin = quad + Q, *x = lin + I, *r = x + I, *p = r + I, *Ap = p + I, *xf = Ap + I;
  __shared__ double s_num_frames, s_totw;
  // OnlineIvectorEstimationStats(ivector_dim, prior_offset, max_count) (ivector-extractor.cc:786-795)
  for (int q = tid; q < Q; q += 256) quad[q] = 0.0;
  for (int j = tid; j < I; j += 256) { lin[j] = 0.0; x[j] = 0.0; }
  __syncthreads();
  if (tid < I) quad[TriIdx(tid, tid)] = 1.0;
  if (tid == 0) { lin[0] = d.prior_offset; s_num_frames = 0.0; }
  __syncthreads();
  const int n_iv = (T + d.period - 1) / d.period;
  for (int i = 0; i < n_iv; i++) {
    const int f0 = i == 0 ? 0 : (i - 1) * d.period + 1, f1 = i * d.period;
    if (tid == 0) s_totw = 0.0;
    // ---- AccStats (ivector-extractor.cc:611-668), pair by pair
    for (int t = f0; t <= f1; t++) {
      for (int k = tid; k < D; k += 256) xf[k] = static_cast<double>(b.raw_lda[(r0 + t) * D + k]);
      __syncthreads();
      for (int j = 0; j < d.ng; j++) {
        const int g = b.post_g[(r0 + t) * d.ng + j];
        if (g < 0) break;                      // uniform: slots are filled from the front
        const double wgt = static_cast<double>(b.post_w[(r0 + t) * d.ng + j]);
        const double *Ug = d.U + static_cast<size_t>(g) * Q;
        for (int q = tid; q < Q; q += 256) quad[q] += wgt * Ug[q];
        if (tid < I) {
          const double *SM = d.SM + static_cast<size_t>(g) * D * I;
          double acc = 0;
          for (int a = 0; a < D; a++) acc += SM[static_cast<size_t>(a) * I + tid] * xf[a];
          lin[tid] += wgt * acc;
        }
        if (tid == 0) s_totw += wgt;
      }
      __syncthreads();
    }
    if (d.max_count > 0.0) {
      const double old_n = s_num_frames, new_n = s_num_frames + s_totw;
      const double change = fmax(new_n, d.max_count) / d.max_count - fmax(old_n, d.max_count) / d.max_count;
      if (change != 0.0) {
        if (tid < I) quad[TriIdx(tid, tid)] += change;
        if (tid == 0) lin[0] += d.prior_offset * change;
      }
    }
    __syncthreads();
    if (tid == 0) s_num_frames += s_totw;
    __syncthreads();
    // ---- GetIvector (ivector-extractor.cc:732-756): LinearCgd from the previous estimate
    if (wave == 0) {
      const int ra = lane, rb = lane + 64;
      const bool ha = ra < I, hb = rb < I;
      if (s_num_frames > 0.0) {
        if (lane == 0 && x[0] == 0.0) x[0] = d.prior_offset;
        __builtin_amdgcn_wave_barrier();
        auto spvec = [&](const double *v, double *ya, double *yb) {     // y = A v for this lane's rows
          double sa = 0, sb = 0;
          for (int c = 0; c < I; c++) {
            const double vc = v[c];
            if (ha) sa += quad[TriIdx(ra, c)] * vc;
            if (hb) sb += quad[TriIdx(rb, c)] * vc;
          }
          *ya = sa; *yb = sb;
        };
        double ya, yb;
        spvec(x, &ya, &yb);
        double pa = 0, pb = 0, rra = 0, rrb = 0;
        if (ha) { pa = lin[ra] - ya; rra = -pa; p[ra] = pa; r[ra] = rra; }
        if (hb) { pb = lin[rb] - yb; rrb = -pb; p[rb] = pb; r[rb] = rrb; }
        double r_cur = WaveSum(rra * rra + rrb * rrb);
        double r_recompute = r_cur;
        const double residual_factor = 1.0e-4, inv_residual_factor = 1.0e4, max_error_sq = 2.2250738585072014e-308;
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < I + 5 && k != d.cg_iters; k++) {
          spvec(p, &ya, &yb);
          if (ha) Ap[ra] = ya;
          if (hb) Ap[rb] = yb;
          const double pr = WaveSum((ha ? p[ra] * r[ra] : 0.0) + (hb ? p[rb] * r[rb] : 0.0));
          const double pAp = WaveSum((ha ? p[ra] * ya : 0.0) + (hb ? p[rb] * yb : 0.0));
          const double alpha = -pr / pAp;
          double na = 0, nb = 0;
          if (ha) { x[ra] += alpha * p[ra]; na = r[ra] + alpha * ya; r[ra] = na; }
          if (hb) { x[rb] += alpha * p[rb]; nb = r[rb] + alpha * yb; r[rb] = nb; }
          double r_next = WaveSum(na * na + nb * nb);
          __builtin_amdgcn_wave_barrier();
          if (r_next < residual_factor * r_recompute || r_next > inv_residual_factor * r_recompute) {
            spvec(x, &ya, &yb);
            if (ha) { na = ya - lin[ra]; r[ra] = na; }
            if (hb) { nb = yb - lin[rb]; r[rb] = nb; }
            r_next = WaveSum(na * na + nb * nb);
            r_recompute = r_next;
          }
          if (r_next <= max_error_sq) break;
          const double beta = r_next / r_cur;
          if (ha) p[ra] = beta * p[ra] - r[ra];
          if (hb) p[rb] = beta * p[rb] - r[rb];
          r_cur = r_next;
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        if (ha) x[ra] = ra == 0 ? d.prior_offset : 0.0;
        if (hb) x[rb] = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      float *o = b.out + (b.out_off[u] + i) * I;
      if (ha) { float v = static_cast<float>(x[ra]); if (ra == 0) v = static_cast<float>(static_cast<double>(v) - d.prior_offset); o[ra] = v; }
      if (hb) o[rb] = static_cast<float>(x[rb]);
    }
    __syncthreads();
  }
}

struct IvExtractor {
  IvDev dev;
  kamd_ivector_desc desc;
  // device copies
  float *d_ldaT = NULL, *d_gconsts = NULL, *d_mivT = NULL, *d_ivT = NULL;
  double *d_gsum = NULL, *d_U = NULL, *d_SM = NULL;
  // workspaces
  double *d_S = NULL; size_t S_cap = 0;
  float *d_nl = NULL, *d_rl = NULL; size_t lda_cap = 0;
  int32_t *d_pg = NULL; float *d_pw = NULL; size_t post_cap = 0;
  int64_t *d_off = NULL; size_t off_cap = 0;
  int64_t last_rows = 0;
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
  if (d.normalize_variance) return bad("variance normalisation (--norm-vars=true) is not supported");
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
  v.cmn_window = d.cmn_window; v.global_frames = d.global_frames; v.normalize_mean = d.normalize_mean;
  v.G = G; v.gconsts = e->d_gconsts; v.mivT = e->d_mivT; v.ivT = e->d_ivT;
  v.I = I; v.Q = Q; v.U = e->d_U; v.SM = e->d_SM;
  v.prior_offset = d.prior_offset; v.max_count = d.max_count;
  v.period = d.ivector_period; v.ng = d.num_gselect; v.cg_iters = d.num_cg_iters;
  v.min_post = d.min_post; v.log_min_post = d.min_post > 0 ? logf(d.min_post) : -INFINITY; v.post_scale = d.posterior_scale;
  return reinterpret_cast<kamd_ivector_extractor *>(e);
}

void kamd_ivector_extractor_destroy(kamd_ivector_extractor *h) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (!e) return;
  void *ps[] = {e->d_ldaT, e->d_gconsts, e->d_mivT, e->d_ivT, e->d_gsum, e->d_U, e->d_SM, e->d_S, e->d_nl, e->d_rl, e->d_pg, e->d_pw, e->d_off};
  for (void *p : ps) if (p) (void)hipFree(p);
  delete e;
}

int kamd_ivector_dim(const kamd_ivector_extractor *h) { return reinterpret_cast<const IvExtractor *>(h)->dev.I; }
int kamd_ivector_period(const kamd_ivector_extractor *h) { return reinterpret_cast<const IvExtractor *>(h)->dev.period; }
int kamd_ivector_num_ivectors(const kamd_ivector_extractor *h, int num_frames) {
  const int P = reinterpret_cast<const IvExtractor *>(h)->dev.period;
  return num_frames <= 0 ? 0 : (num_frames + P - 1) / P;
}

int kamd_ivector_extract_online_device(kamd_ivector_extractor *h, const float *d_feats, const int64_t *h_row_off, int ld_feat,
                                       int n_utts, float *d_out, const int64_t *h_out_row_off, void *stream) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_utts <= 0) return KAMD_OK;
  const kamd::IvDev &v = e->dev;
  if (ld_feat < v.feat_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_feat %d < feature dim %d", ld_feat, v.feat_dim);
  int max_T = 0;
  for (int u = 0; u < n_utts; u++) {
    const int64_t T = h_row_off[u + 1] - h_row_off[u];
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames", u);
    if (h_out_row_off[u + 1] - h_out_row_off[u] < (T + v.period - 1) / v.period)
      return kamd::SetError(KAMD_ERR_ARG, "utterance %d: output rows too few for %lld frames", u, static_cast<long long>(T));
    max_T = std::max<int>(max_T, static_cast<int>(T));
  }
  const int64_t rows = h_row_off[n_utts] - h_row_off[0];
  if (kamd::GrowDev(&e->d_S, &e->S_cap, static_cast<size_t>(rows) * v.feat_dim) != KAMD_OK) return KAMD_ERR_HIP;
  size_t cap2 = e->lda_cap;
  if (kamd::GrowDev(&e->d_nl, &e->lda_cap, static_cast<size_t>(rows) * v.D) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_rl, &cap2, static_cast<size_t>(rows) * v.D) != KAMD_OK) return KAMD_ERR_HIP;
  size_t cap3 = e->post_cap;
  if (kamd::GrowDev(&e->d_pg, &e->post_cap, static_cast<size_t>(rows) * v.ng) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_pw, &cap3, static_cast<size_t>(rows) * v.ng) != KAMD_OK) return KAMD_ERR_HIP;
  if (kamd::GrowDev(&e->d_off, &e->off_cap, static_cast<size_t>(2 * (n_utts + 1))) != KAMD_OK) return KAMD_ERR_HIP;
  KAMD_HIP(hipMemcpyAsync(e->d_off, h_row_off, (n_utts + 1) * 8, hipMemcpyHostToDevice, st));
  KAMD_HIP(hipMemcpyAsync(e->d_off + n_utts + 1, h_out_row_off, (n_utts + 1) * 8, hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));          // the offset arrays are pageable host memory
  kamd::IvBatch b;
  b.feats = d_feats; b.ld = ld_feat; b.row_off = e->d_off; b.out_off = e->d_off + n_utts + 1;
  b.S = e->d_S; b.norm_lda = e->d_nl; b.raw_lda = e->d_rl; b.post_g = e->d_pg; b.post_w = e->d_pw; b.out = d_out;
  b.row_base = h_row_off[0];
  hipLaunchKernelGGL(kamd::PrefixKernel, dim3(n_utts), dim3(256), 0, st, v, b);
  const size_t lds_front = static_cast<size_t>(2) * (kamd::IV_FT + v.L + v.R) * v.feat_dim * sizeof(float);
  hipLaunchKernelGGL(kamd::FrontKernel, dim3(kamd::CeilDiv(max_T, kamd::IV_FT), n_utts), dim3(256), lds_front, st, v, b);
  const size_t lds_post = static_cast<size_t>(4) * (v.D + v.G) * sizeof(float);
  hipLaunchKernelGGL(kamd::PostKernel, dim3(kamd::CeilDiv(rows, 4)), dim3(256), lds_post, st, v, b, rows);
  const size_t lds_solve = (static_cast<size_t>(v.Q) + 5 * v.I + v.D) * sizeof(double);
  hipLaunchKernelGGL(kamd::SolveKernel, dim3(n_utts), dim3(256), lds_solve, st, v, b);
  KAMD_HIP(hipGetLastError());
  e->last_rows = rows;
  return KAMD_OK;
}

int kamd_ivector_extract_online(kamd_ivector_extractor *h, const float *feats, int num_frames, float *out, int out_rows_cap) {
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
  if (rc == KAMD_OK) rc = kamd_ivector_extract_online_device(h, d_f, ro, dim, 1, d_o, oo, NULL);
  if (rc == KAMD_OK && hipMemcpy(out, d_o, static_cast<size_t>(n) * I * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "i-vector extraction failed: %s", hipGetErrorString(hipGetLastError()));
  (void)hipFree(d_f); (void)hipFree(d_o);
  return rc == KAMD_OK ? n : rc;
}

int kamd_ivector_last_posteriors(kamd_ivector_extractor *h, int32_t *gauss, float *weight, int64_t frames_cap) {
  IvExtractor *e = reinterpret_cast<IvExtractor *>(h);
  if (frames_cap < e->last_rows) return kamd::SetError(KAMD_ERR_ARG, "buffer too small for %lld frames", static_cast<long long>(e->last_rows));
  KAMD_HIP(hipMemcpy(gauss, e->d_pg, static_cast<size_t>(e->last_rows) * e->dev.ng * 4, hipMemcpyDeviceToHost));
  KAMD_HIP(hipMemcpy(weight, e->d_pw, static_cast<size_t>(e->last_rows) * e->dev.ng * 4, hipMemcpyDeviceToHost));
  return KAMD_OK;
}

}  // extern "C"
