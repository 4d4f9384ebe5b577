// What v_mfma_f32_32x32x2_f32 sustains on this chip with nothing else going on: every wavefront runs a long chain of
// independent MFMAs on register operands (no LDS, no memory).  Reports TFLOP/s and the shader clock implied by the
// wall-clock counter, at 1, 2, 3 and 4 wavefronts per SIMD.   build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void MfmaLoop(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  float *d_out;
  hipMalloc(&d_out, sizeof(float) * cus * 16 * 256);
  const int iters = 200000;
  for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu++) {          // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(MfmaLoop<4>, dim3(cus * wg_per_cu), dim3(256), 0, 0, d_out, 1000, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(MfmaLoop<4>, dim3(cus * wg_per_cu), dim3(256), 0, 0, d_out, iters, 1.0f, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 2 * 4.0 * iters * 4.0 * wg_per_cu * cus;   // per MFMA x NACC x iters x waves
    const double mfma_cycles = 64.0 * 4.0 * iters * wg_per_cu;                       // per SIMD, if back to back
    printf("%d CUs, %d wave(s) per SIMD: %.2f ms, %.1f TFLOP/s; back-to-back issue would need %.2f GHz\n", cus, wg_per_cu, ms,
           flops / (ms * 1e-3) / 1e12, mfma_cycles / (ms * 1e-3) / 1e9);
  }
  // sustained: the same loop for ~0.1 s to ~1.5 s at two waves per SIMD (power management has time to act)
  for (int scale = 1; scale <= 16; scale *= 4) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(MfmaLoop<4>, dim3(cus * 2), dim3(256), 0, 0, d_out, iters * 2 * scale, 1.0f, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 2 * 4.0 * iters * 2 * scale * 4.0 * 2 * cus;
    printf("sustained, 2 waves per SIMD, %.0f ms: %.1f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
