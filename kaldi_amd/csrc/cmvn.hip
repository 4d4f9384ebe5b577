// cmvn.hip -- compute-cmvn-stats / apply-cmvn on the device (transform/cmvn.cc:30-118), per utterance
// of a batch whose features are already in HBM.  Elementwise, HBM-trivial: one pass to accumulate
// [sum, sum of squares, count] in fp64, one pass x <- x * scale + offset.
#include <vector>

#include "common.h"

namespace kamd {

// stats[u] = [2][dim + 1] doubles: row 0 sums and (last column) the count, row 1 sums of squares
// weights (optional, one per row of the batch): AccCmvnStats(frame, weight, stats) for frames of non-zero weight (:49-62)
__global__ __launch_bounds__(256) void CmvnStatsKernel(const float *feats, int ld, int dim, const int64_t *row_off, const float *weights,
                                                       double *stats) {
  __shared__ double s1[256], s2[256], s3[256];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int64_t r0 = row_off[u];
  const int T = static_cast<int>(row_off[u + 1] - r0);
  int dpw = 1;
  while (dpw < dim) dpw <<= 1;
  const int chunks = 256 / dpw, k = tid % dpw, c = tid / dpw;
  double a = 0, b = 0, cnt = 0;
  if (k < dim)
    for (int t = c; t < T; t += chunks) {
      const float x = feats[(r0 + t) * ld + k];
      const float w = weights ? weights[r0 + t] : 1.0f;
      if (w != 0.0f) {
        a += static_cast<double>(x * w);             // AccCmvnStats: *mean_ptr += *feats_ptr * weight (float product)
        b += static_cast<double>(x * x * w);         //               *var_ptr += *feats_ptr * *feats_ptr * weight
        cnt += static_cast<double>(w);               //               *count_ptr += weight
      }
    }
  s1[tid] = a; s2[tid] = b; s3[tid] = cnt;
  __syncthreads();
  if (c == 0 && k < dim) {
    for (int c2 = 1; c2 < chunks; c2++) { a += s1[c2 * dpw + k]; b += s2[c2 * dpw + k]; cnt += s3[c2 * dpw + k]; }
    double *o = stats + static_cast<size_t>(u) * 2 * (dim + 1);
    o[k] += a; o[dim + 1 + k] += b;
    if (k == 0) o[dim] += cnt;
  }
}

// ApplyCmvn (transform/cmvn.cc:64-118): means only: x + float(-sum / count); with variances:
// x * float(1 / sqrt(var)) + float(-mean / sqrt(var)), var floored at 1e-20
// reverse: ApplyCmvnReverse (:120-168): x * float(sqrt(var)) + float(mean) (zero-mean unit-variance data -> the statistics')
__global__ __launch_bounds__(256) void CmvnApplyKernel(float *feats, int ld, int dim, const int64_t *row_off, const double *stats,
                                                       int norm_vars, int reverse, int *bad) {
  extern __shared__ float norm[];                  // offset[dim], scale[dim]
  const int u = blockIdx.y, tid = threadIdx.x;
  const int64_t r0 = row_off[u];
  const int T = static_cast<int>(row_off[u + 1] - r0);
  const double *st = stats + static_cast<size_t>(u) * 2 * (dim + 1);
  const double count = st[dim];
  if (count < 1.0) { if (tid == 0 && blockIdx.x == 0) atomicExch(bad, u + 1); return; }   // "Insufficient stats"
  for (int k = tid; k < dim; k += 256) {
    if (reverse) {
      const double mean = st[k] / count;
      double scale = 1.0;
      if (norm_vars) {
        double var = st[dim + 1 + k] / count - mean * mean;
        if (var < 1.0e-20) var = 1.0e-20;
        scale = sqrt(var);
      }
      norm[k] = static_cast<float>(mean); norm[dim + k] = static_cast<float>(scale);
    } else if (!norm_vars) { norm[k] = static_cast<float>(-1.0 / count * st[k]); norm[dim + k] = 1.0f; }
    else {
      const double mean = st[k] / count;
      double var = st[dim + 1 + k] / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      const double scale = 1.0 / sqrt(var);
      norm[k] = static_cast<float>(-(mean * scale)); norm[dim + k] = static_cast<float>(scale);
    }
  }
  __syncthreads();
  const int rows_per_block = 64;
  const int t0 = blockIdx.x * rows_per_block, t1 = min(T, t0 + rows_per_block);
  for (int i = tid; i < (t1 - t0) * dim; i += 256) {
    const int t = t0 + i / dim, k = i % dim;
    float x = feats[(r0 + t) * ld + k];
    if (norm_vars) x = x * norm[dim + k];          // MulColsVec, then AddVecToRows: two roundings
    feats[(r0 + t) * ld + k] = x + norm[k];
  }
}

// add-deltas: DeltaFeatures::Process (feat/feature-functions.cc:142-165) for every frame of every utterance;
// scales = the (order + 1) windows of DeltaFeatures::DeltaFeatures (:118-140), concatenated, scale_off[i] their starts
__global__ __launch_bounds__(256) void AddDeltasKernel(const float *in, int ld_in, float *out, int ld_out, const int64_t *row_off, int dim, int order,
                                                       const float *scales, const int *scale_off) {
  const int u = blockIdx.y;
  const int64_t r0 = row_off[u];
  const int T = static_cast<int>(row_off[u + 1] - r0);
  const int n = (order + 1) * dim;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; idx < static_cast<int64_t>(T) * n; idx += static_cast<int64_t>(gridDim.x) * 256) {
    const int t = static_cast<int>(idx / n), c = static_cast<int>(idx - static_cast<int64_t>(t) * n), i = c / dim, k = c - i * dim;
    const float *sc = scales + scale_off[i];
    const int max_offset = (scale_off[i + 1] - scale_off[i] - 1) / 2;
    float acc = 0.f;
    for (int j = -max_offset; j <= max_offset; j++) {
      int f = t + j;
      f = f < 0 ? 0 : (f >= T ? T - 1 : f);
      const float s = sc[j + max_offset];
      if (s != 0.0f) acc = acc + s * in[(r0 + f) * ld_in + k];
    }
    out[(r0 + t) * ld_out + c] = acc;
  }
}

// splice-feats | transform-feats: SpliceFrames (feat/feature-functions.cc:205-231, context clamped at the
// utterance's ends) followed by y = M x (+ offset column when M has one more column), transform-feats.cc:120-150.
// transform == NULL: the spliced vectors themselves.  utt_xf[u]: which of the uploaded transforms utterance u uses.
__global__ __launch_bounds__(256) void SpliceTransformKernel(const float *in, int ld_in, float *out, int ld_out, const int64_t *row_off, int dim,
                                                             int left, int right, const float *xf, const int *utt_xf, int xf_rows, int xf_cols) {
  const int u = blockIdx.y;
  const int64_t r0 = row_off[u];
  const int T = static_cast<int>(row_off[u + 1] - r0);
  const int sd = dim * (left + 1 + right), n = xf ? xf_rows : sd;
  const float *M = xf ? xf + static_cast<size_t>(utt_xf[u]) * xf_rows * xf_cols : NULL;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; idx < static_cast<int64_t>(T) * n; idx += static_cast<int64_t>(gridDim.x) * 256) {
    const int t = static_cast<int>(idx / n), o = static_cast<int>(idx - static_cast<int64_t>(t) * n);
    float acc;
    if (!M) {
      int t2 = t - left + o / dim;
      t2 = t2 < 0 ? 0 : (t2 >= T ? T - 1 : t2);
      acc = in[(r0 + t2) * ld_in + o % dim];
    } else {
      acc = xf_cols == sd + 1 ? M[static_cast<size_t>(o) * xf_cols + sd] : 0.f;
      for (int c = 0; c <= left + right; c++) {
        int t2 = t - left + c;
        t2 = t2 < 0 ? 0 : (t2 >= T ? T - 1 : t2);
        const float *x = in + (r0 + t2) * ld_in, *m = M + static_cast<size_t>(o) * xf_cols + c * dim;
        for (int k = 0; k < dim; k++) acc = acc + m[k] * x[k];
      }
    }
    out[(r0 + t) * ld_out + o] = acc;
  }
}

}  // namespace kamd

extern "C" {

int kamd_feat_splice_transform_device(const float *d_in, int ld_in, float *d_out, int ld_out, const int64_t *h_row_off, int n_utts, int dim,
                                      int left, int right, const float *h_transforms, int n_transforms, const int32_t *h_utt_transform,
                                      int xf_rows, int xf_cols, void *stream) {
  if (!kamd::RequireDevice()) return KAMD_ERR_HIP;
  if (n_utts <= 0) return KAMD_OK;
  const int sd = dim * (left + 1 + right);
  if (dim <= 0 || left < 0 || right < 0 || ld_in < dim) return kamd::SetError(KAMD_ERR_ARG, "splice / transform: bad dimensions");
  if (h_transforms) {
    if (n_transforms <= 0 || xf_rows <= 0 || (xf_cols != sd && xf_cols != sd + 1))
      return kamd::SetError(KAMD_ERR_ARG, "Transform matrix has bad dimension %dx%d versus feat dim %d", xf_rows, xf_cols, sd);   // transform-feats.cc:139-145
    for (int u = 0; u < n_utts; u++)
      if (h_utt_transform[u] < 0 || h_utt_transform[u] >= n_transforms) return kamd::SetError(KAMD_ERR_ARG, "utterance %d: no such transform", u);
  }
  const int n_out = h_transforms ? xf_rows : sd;
  if (ld_out < n_out) return kamd::SetError(KAMD_ERR_ARG, "splice / transform: output leading dimension too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *d_xf = NULL; int *d_ux = NULL; int64_t *d_ro = NULL;
  int rc = KAMD_OK, max_T = 0;
  for (int u = 0; u < n_utts; u++) max_T = std::max<int>(max_T, static_cast<int>(h_row_off[u + 1] - h_row_off[u]));
  if (hipMalloc(reinterpret_cast<void **>(&d_ro), (n_utts + 1) * sizeof(int64_t)) != hipSuccess) return kamd::SetError(KAMD_ERR_HIP, "allocation failed");
  if (hipMemcpyAsync(d_ro, h_row_off, (n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "upload failed");
  if (rc == KAMD_OK && h_transforms) {
    const size_t n = static_cast<size_t>(n_transforms) * xf_rows * xf_cols;
    if (hipMalloc(reinterpret_cast<void **>(&d_xf), n * sizeof(float)) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d_ux), n_utts * sizeof(int)) != hipSuccess ||
        hipMemcpyAsync(d_xf, h_transforms, n * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(d_ux, h_utt_transform, n_utts * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
      rc = kamd::SetError(KAMD_ERR_HIP, "transform upload failed");
  }
  if (rc == KAMD_OK && max_T > 0) {
    const int64_t work = static_cast<int64_t>(max_T) * n_out;
    hipLaunchKernelGGL(kamd::SpliceTransformKernel, dim3(static_cast<unsigned>(std::min<int64_t>((work + 255) / 256, 4096)), n_utts), dim3(256), 0, st, d_in,
                       ld_in, d_out, ld_out, d_ro, dim, left, right, d_xf, d_ux, xf_rows, xf_cols);
    if (hipStreamSynchronize(st) != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "splice / transform kernel failed: %s", hipGetErrorString(hipGetLastError()));
  }
  if (d_xf) (void)hipFree(d_xf);
  if (d_ux) (void)hipFree(d_ux);
  (void)hipFree(d_ro);
  return rc;
}

int kamd_feat_add_deltas_device(const float *d_in, int ld_in, float *d_out, int ld_out, const int64_t *h_row_off, int n_utts, int dim, int order,
                                int window, void *stream) {
  if (!kamd::RequireDevice()) return KAMD_ERR_HIP;
  if (n_utts <= 0) return KAMD_OK;
  if (order < 0 || order >= 1000 || window <= 0 || window >= 1000 || dim <= 0 || ld_in < dim || ld_out < (order + 1) * dim)
    return kamd::SetError(KAMD_ERR_ARG, "add-deltas: bad order / window / dimensions");
  // DeltaFeatures::DeltaFeatures: scales_[i] = scales_[i-1] convolved with j / sum(j^2), j = -window .. window (BaseFloat arithmetic)
  std::vector<std::vector<float> > sc(order + 1);
  sc[0].assign(1, 1.0f);
  for (int i = 1; i <= order; i++) {
    const std::vector<float> &prev = sc[i - 1];
    std::vector<float> &cur = sc[i];
    const int prev_offset = (static_cast<int>(prev.size()) - 1) / 2, cur_offset = prev_offset + window;
    cur.assign(prev.size() + 2 * window, 0.0f);
    float normalizer = 0.0f;
    for (int j = -window; j <= window; j++) {
      normalizer += j * j;
      for (int k = -prev_offset; k <= prev_offset; k++) cur[j + k + cur_offset] += static_cast<float>(j) * prev[k + prev_offset];
    }
    for (float &v : cur) v *= 1.0f / normalizer;
  }
  std::vector<float> flat; std::vector<int> off(1, 0);
  for (int i = 0; i <= order; i++) { flat.insert(flat.end(), sc[i].begin(), sc[i].end()); off.push_back(static_cast<int>(flat.size())); }
  hipStream_t st = static_cast<hipStream_t>(stream);
  float *d_sc = NULL; int *d_off = NULL; int64_t *d_ro = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_sc), flat.size() * sizeof(float)));
  if (hipMalloc(reinterpret_cast<void **>(&d_off), off.size() * sizeof(int)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&d_ro), (n_utts + 1) * sizeof(int64_t)) != hipSuccess) {
    (void)hipFree(d_sc); if (d_off) (void)hipFree(d_off);
    return kamd::SetError(KAMD_ERR_HIP, "allocation failed");
  }
  int rc = KAMD_OK, max_T = 0;
  for (int u = 0; u < n_utts; u++) max_T = std::max<int>(max_T, static_cast<int>(h_row_off[u + 1] - h_row_off[u]));
  if (hipMemcpyAsync(d_sc, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_ro, h_row_off, (n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "add-deltas: upload failed");
  if (rc == KAMD_OK && max_T > 0) {
    const int64_t work = static_cast<int64_t>(max_T) * (order + 1) * dim;
    hipLaunchKernelGGL(kamd::AddDeltasKernel, dim3(static_cast<unsigned>(std::min<int64_t>((work + 255) / 256, 4096)), n_utts), dim3(256), 0, st, d_in, ld_in,
                       d_out, ld_out, d_ro, dim, order, d_sc, d_off);
    if (hipStreamSynchronize(st) != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "add-deltas kernel failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(d_sc); (void)hipFree(d_off); (void)hipFree(d_ro);
  return rc;
}

static int CmvnAccStats(const float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, const float *d_weights, double *h_stats,
                        void *stream) {
  if (!kamd::RequireDevice()) return KAMD_ERR_HIP;
  if (n_utts <= 0) return KAMD_OK;
  if (dim <= 0 || dim > 256 || ld < dim) return kamd::SetError(KAMD_ERR_ARG, "cmvn: bad feature dim / leading dimension");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = static_cast<size_t>(n_utts) * 2 * (dim + 1);
  double *d_stats = NULL; int64_t *d_off = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_stats), n * sizeof(double)));
  if (hipMalloc(reinterpret_cast<void **>(&d_off), (n_utts + 1) * sizeof(int64_t)) != hipSuccess) { (void)hipFree(d_stats); return kamd::SetError(KAMD_ERR_HIP, "allocation failed"); }
  int rc = KAMD_OK;
  if (hipMemcpyAsync(d_stats, h_stats, n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_off, h_row_off, (n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "cmvn: upload failed");
  if (rc == KAMD_OK) {
    hipLaunchKernelGGL(kamd::CmvnStatsKernel, dim3(n_utts), dim3(256), 0, st, d_feats, ld, dim, d_off, d_weights, d_stats);
    if (hipMemcpyAsync(h_stats, d_stats, n * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = kamd::SetError(KAMD_ERR_HIP, "cmvn: statistics kernel failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(d_stats); (void)hipFree(d_off);
  return rc;
}

int kamd_cmvn_acc_stats_device(const float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, double *h_stats,
                               void *stream) {
  return CmvnAccStats(d_feats, h_row_off, ld, dim, n_utts, NULL, h_stats, stream);
}
int kamd_cmvn_acc_stats_weighted_device(const float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, const float *d_weights,
                                        double *h_stats, void *stream) {
  if (!d_weights) return kamd::SetError(KAMD_ERR_ARG, "cmvn: null weights");
  return CmvnAccStats(d_feats, h_row_off, ld, dim, n_utts, d_weights, h_stats, stream);
}

static int CmvnApply(float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, const double *h_stats,
                     int norm_means, int norm_vars, int reverse, void *stream) {
  if (!kamd::RequireDevice()) return KAMD_ERR_HIP;
  if (n_utts <= 0) return KAMD_OK;
  if (dim <= 0 || ld < dim) return kamd::SetError(KAMD_ERR_ARG, "cmvn: bad feature dim / leading dimension");
  if (norm_vars && !norm_means) return kamd::SetError(KAMD_ERR_ARG, "You cannot normalize the variance but not the mean.");   // apply-cmvn.cc:63-64
  if (!norm_means) return KAMD_OK;                 // apply-cmvn copies the features unchanged
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = static_cast<size_t>(n_utts) * 2 * (dim + 1);
  double *d_stats = NULL; int64_t *d_off = NULL; int *d_bad = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_stats), n * sizeof(double) + 16));
  if (hipMalloc(reinterpret_cast<void **>(&d_off), (n_utts + 1) * sizeof(int64_t)) != hipSuccess) { (void)hipFree(d_stats); return kamd::SetError(KAMD_ERR_HIP, "allocation failed"); }
  d_bad = reinterpret_cast<int *>(d_stats + n);
  int rc = KAMD_OK, bad = 0, max_T = 0;
  for (int u = 0; u < n_utts; u++) max_T = std::max<int>(max_T, static_cast<int>(h_row_off[u + 1] - h_row_off[u]));
  if (hipMemcpyAsync(d_stats, h_stats, n * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemsetAsync(d_bad, 0, 4, st) != hipSuccess ||
      hipMemcpyAsync(d_off, h_row_off, (n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "cmvn: upload failed");
  if (rc == KAMD_OK && max_T > 0) {
    hipLaunchKernelGGL(kamd::CmvnApplyKernel, dim3(kamd::CeilDiv(max_T, 64), n_utts), dim3(256), 2 * dim * sizeof(float), st, d_feats, ld, dim,
                       d_off, d_stats, norm_vars, reverse, d_bad);
    if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = kamd::SetError(KAMD_ERR_HIP, "cmvn: apply kernel failed: %s", hipGetErrorString(hipGetLastError()));
  }
  (void)hipFree(d_stats); (void)hipFree(d_off);
  if (rc == KAMD_OK && bad)
    return kamd::SetError(KAMD_ERR_ARG, "Insufficient stats for cepstral mean and variance normalization: utterance %d, count < 1", bad - 1);
  return rc;
}

int kamd_cmvn_apply_device(float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, const double *h_stats,
                           int norm_means, int norm_vars, void *stream) {
  return CmvnApply(d_feats, h_row_off, ld, dim, n_utts, h_stats, norm_means, norm_vars, 0, stream);
}
int kamd_cmvn_apply_reverse_device(float *d_feats, const int64_t *h_row_off, int ld, int dim, int n_utts, const double *h_stats,
                                   int norm_means, int norm_vars, void *stream) {
  return CmvnApply(d_feats, h_row_off, ld, dim, n_utts, h_stats, norm_means, norm_vars, 1, stream);
}

}  // extern "C"
