// gmm.hip -- DecodableAmDiagGmmScaled's numbers for a whole utterance batch (gmm/decodable-am-diag-gmm.cc:27-70,
// gmm/diag-gmm.cc:520-545): out[t][pdf] = scale * LogSumExp_m( gconst_m + means_invvars_m . x_t - 0.5 inv_vars_m . x_t^2 ).
// BASELINE configs[0] (egs/yesno monophone GMM + gmm-latgen-faster) is the reference's CPU plumbing case; here the
// same decodable feeds the device decoder.  One thread per (frame, pdf), the frame's features staged in LDS, the
// mixture parameters transposed [dim][gauss] so that consecutive pdfs' first components are read coalesced-ish;
// trivial next to the search.
#include <vector>

#include "common.h"

namespace kamd {

struct AmGmm {
  int num_pdfs = 0, dim = 0, num_gauss = 0;
  int *d_mix_off = NULL;            // [num_pdfs + 1]
  float *d_gconsts = NULL;          // [num_gauss]
  float *d_miv = NULL, *d_iv = NULL;   // [num_gauss][dim], inv_vars premultiplied by -0.5
};

constexpr int GMM_FT = 4;           // frames per workgroup

__global__ __launch_bounds__(256) void GmmLogLikesKernel(int num_pdfs, int dim, const int *mix_off, const float *gconsts, const float *miv,
                                                         const float *iv, const float *feats, int ld, int64_t rows, float scale, float *out) {
  extern __shared__ float xs[];     // [GMM_FT][2 * dim]: x, x^2
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * GMM_FT;
  for (int i = threadIdx.x; i < GMM_FT * dim; i += 256) {
    const int f = i / dim, k = i - f * dim;
    const float v = t0 + f < rows ? feats[(t0 + f) * ld + k] : 0.f;
    xs[f * 2 * dim + k] = v; xs[f * 2 * dim + dim + k] = v * v;
  }
  __syncthreads();
  for (int p = blockIdx.y * 256 + threadIdx.x; p < num_pdfs; p += gridDim.y * 256) {
    const int g0 = mix_off[p], g1 = mix_off[p + 1];
    for (int f = 0; f < GMM_FT && t0 + f < rows; f++) {
      const float *x = xs + f * 2 * dim, *x2 = x + dim;
      float mx = -INFINITY;
      for (int g = g0; g < g1; g++) {                       // pass 1: the maximum (VectorBase::Max)
        float acc = gconsts[g];
        const float *m = miv + static_cast<size_t>(g) * dim, *v = iv + static_cast<size_t>(g) * dim;
        for (int k = 0; k < dim; k++) acc = acc + m[k] * x[k];
        for (int k = 0; k < dim; k++) acc = acc + v[k] * x2[k];
        mx = fmaxf(mx, acc);
      }
      // VectorBase<float>::LogSumExp(prune = -1) (matrix/kaldi-vector.cc:760-778): terms below max + log(FLT_EPSILON) dropped
      const float cutoff = mx + (-15.9423847198486328125f);
      double sum = 0.0;
      for (int g = g0; g < g1; g++) {
        float acc = gconsts[g];
        const float *m = miv + static_cast<size_t>(g) * dim, *v = iv + static_cast<size_t>(g) * dim;
        for (int k = 0; k < dim; k++) acc = acc + m[k] * x[k];
        for (int k = 0; k < dim; k++) acc = acc + v[k] * x2[k];
        if (acc >= cutoff) sum += static_cast<double>(expf(acc - mx));
      }
      const float log_sum = mx + static_cast<float>(log(sum));
      out[(t0 + f) * num_pdfs + p] = scale * log_sum;
    }
  }
}

}  // namespace kamd
using kamd::AmGmm;

extern "C" {

kamd_am_gmm *kamd_am_gmm_create(int32_t num_pdfs, int32_t dim, const int32_t *mix_off, const float *gconsts, const float *means_invvars,
                                const float *inv_vars) {
  if (!kamd::RequireDevice()) return NULL;
  if (num_pdfs <= 0 || dim <= 0 || dim > 4096 || mix_off[0] != 0) { kamd::SetError(KAMD_ERR_ARG, "AmDiagGmm: bad sizes"); return NULL; }
  for (int p = 0; p < num_pdfs; p++)
    if (mix_off[p + 1] <= mix_off[p]) { kamd::SetError(KAMD_ERR_ARG, "AmDiagGmm: pdf %d has no Gaussians", p); return NULL; }
  AmGmm *g = new AmGmm();
  g->num_pdfs = num_pdfs; g->dim = dim; g->num_gauss = mix_off[num_pdfs];
  const size_t n = static_cast<size_t>(g->num_gauss) * dim;
  std::vector<float> half(n);
  for (size_t i = 0; i < n; i++) half[i] = -0.5f * inv_vars[i];
  bool ok = hipMalloc(reinterpret_cast<void **>(&g->d_mix_off), (num_pdfs + 1) * sizeof(int)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&g->d_gconsts), g->num_gauss * sizeof(float)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&g->d_miv), n * sizeof(float)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&g->d_iv), n * sizeof(float)) == hipSuccess;
  ok = ok && hipMemcpy(g->d_mix_off, mix_off, (num_pdfs + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(g->d_gconsts, gconsts, g->num_gauss * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(g->d_miv, means_invvars, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(g->d_iv, half.data(), n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { kamd::SetError(KAMD_ERR_HIP, "AmDiagGmm: device allocation failed"); kamd_am_gmm_destroy(reinterpret_cast<kamd_am_gmm *>(g)); return NULL; }
  return reinterpret_cast<kamd_am_gmm *>(g);
}

void kamd_am_gmm_destroy(kamd_am_gmm *h) {
  AmGmm *g = reinterpret_cast<AmGmm *>(h);
  if (!g) return;
  void *ps[] = {g->d_mix_off, g->d_gconsts, g->d_miv, g->d_iv};
  for (void *p : ps) if (p) (void)hipFree(p);
  delete g;
}

int kamd_am_gmm_num_pdfs(const kamd_am_gmm *h) { return reinterpret_cast<const AmGmm *>(h)->num_pdfs; }
int kamd_am_gmm_dim(const kamd_am_gmm *h) { return reinterpret_cast<const AmGmm *>(h)->dim; }

int kamd_am_gmm_loglikes_device(kamd_am_gmm *h, const float *d_feats, int ld, int64_t rows, float scale, float *d_out, void *stream) {
  AmGmm *g = reinterpret_cast<AmGmm *>(h);
  if (rows <= 0) return KAMD_OK;
  if (ld < g->dim) return kamd::SetError(KAMD_ERR_ARG, "Dim mismatch: data dim = %d vs. model dim = %d", ld, g->dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(kamd::CeilDiv(rows, kamd::GMM_FT), std::min(kamd::CeilDiv(g->num_pdfs, 256), 64));
  hipLaunchKernelGGL(kamd::GmmLogLikesKernel, grid, dim3(256), static_cast<size_t>(kamd::GMM_FT) * 2 * g->dim * sizeof(float), st, g->num_pdfs,
                     g->dim, g->d_mix_off, g->d_gconsts, g->d_miv, g->d_iv, d_feats, ld, rows, scale, d_out);
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}

int kamd_am_gmm_loglikes(kamd_am_gmm *h, const float *feats, int num_frames, int feat_dim, float scale, float *out) {
  AmGmm *g = reinterpret_cast<AmGmm *>(h);
  if (num_frames <= 0) return KAMD_OK;
  if (feat_dim != g->dim) return kamd::SetError(KAMD_ERR_ARG, "Dim mismatch: data dim = %d vs. model dim = %d", feat_dim, g->dim);
  float *d_f = NULL, *d_o = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_f), static_cast<size_t>(num_frames) * feat_dim * sizeof(float)));
  if (hipMalloc(reinterpret_cast<void **>(&d_o), static_cast<size_t>(num_frames) * g->num_pdfs * sizeof(float)) != hipSuccess) {
    (void)hipFree(d_f);
    return kamd::SetError(KAMD_ERR_HIP, "allocation failed");
  }
  int rc = KAMD_OK;
  if (hipMemcpy(d_f, feats, static_cast<size_t>(num_frames) * feat_dim * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "upload failed");
  if (rc == KAMD_OK) rc = kamd_am_gmm_loglikes_device(h, d_f, feat_dim, num_frames, scale, d_o, NULL);
  if (rc == KAMD_OK && hipMemcpy(out, d_o, static_cast<size_t>(num_frames) * g->num_pdfs * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "GMM log-likelihoods failed: %s", hipGetErrorString(hipGetLastError()));
  (void)hipFree(d_f); (void)hipFree(d_o);
  return rc;
}

}  // extern "C"
