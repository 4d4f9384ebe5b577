// synth.hip -- benchmark workload synthesis on the device (not part of the decode path): a planted log-likelihood matrix
// for the bench's --planted variant.  Row r gets noise * N(0, 1) on every pdf and + peak on the pdf the planted path
// visits on that frame -- kaldi_amd/synth.py sample_utterance's recipe without its per-row normalisation (a per-frame
// constant moves every token of the frame alike and changes nothing in the search) -- from a counter-based generator,
// so that 2620 utterances x 6000 pdfs (15 GB) need neither a host pass nor another library.
#include "common.h"

namespace kamd {

__device__ inline unsigned Mix(unsigned long long x) {          // splitmix64 finaliser, upper half
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
  return static_cast<unsigned>(x >> 32);
}

__global__ __launch_bounds__(256) void PlantedLoglikesKernel(float *out, int64_t rows, int P, int ld, const int32_t *true_pdf, float peak,
                                                             float noise, unsigned long long seed) {
  const int64_t pairs_per_row = (P + 1) / 2;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < rows * pairs_per_row;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = i / pairs_per_row;
    const int p = static_cast<int>(i - r * pairs_per_row) * 2;
    const unsigned long long c = seed + 0x9e3779b97f4a7c15ULL * static_cast<unsigned long long>(i + 1);
    const float u1 = (Mix(c) + 1.0f) * (1.0f / 4294967808.0f), u2 = Mix(c ^ 0xd1b54a32d192ed03ULL) * (1.0f / 4294967296.0f);
    const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.2831853f * u2;          // Box-Muller: two normals per pair of pdfs
    const int t = true_pdf[r];
    float *row = out + r * ld;
    row[p] = noise * rad * cosf(ang) + (p == t ? peak : 0.0f);
    if (p + 1 < P) row[p + 1] = noise * rad * sinf(ang) + (p + 1 == t ? peak : 0.0f);
  }
}

}  // namespace kamd

extern "C" int kamd_synth_planted_loglikes_device(float *d_out, int64_t rows, int num_pdfs, int ld, const int32_t *d_true_pdf, float peak,
                                                  float noise, uint64_t seed, void *stream) {
  if (rows <= 0) return KAMD_OK;
  if (!d_out || !d_true_pdf || num_pdfs <= 0 || ld < num_pdfs) return kamd::SetError(KAMD_ERR_ARG, "planted log-likelihoods: bad arguments");
  hipLaunchKernelGGL(kamd::PlantedLoglikesKernel, dim3(8192), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, rows, num_pdfs, ld,
                     d_true_pdf, peak, noise, static_cast<unsigned long long>(seed));
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}
