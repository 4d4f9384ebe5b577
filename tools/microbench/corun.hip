// Does a grid kernel slow down out of proportion when a few persistent 1024-thread workgroups hold CUs beside it?
// (Round 2: 16 decoder lanes beside the acoustic model's GEMMs made those 1.5x slower -- 16 of 256 CUs.)
//   A = a stream of `launches` grid kernels: 256-thread workgroups of register-only MFMA work (a GEMM's occupancy: 3 per CU)
//   B = N persistent workgroups of 1024 threads x 128 VGPRs + 96 KB LDS (a decoder lane's footprint: a CU each), which
//       spin on a host flag (mode 0), or stream writes + reads over their own 64 MB (mode 1), or do scattered 4-byte
//       atomics on it (mode 2)
// Reported: time of A alone, and beside N = 16 / 64 occupiers of each kind; "proportional" = alone * 256 / (256 - N).
// build: hipcc --offload-arch=gfx950 -O3 -o corun corun.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void A(int iters, float *out, const float4 *src, int mem_every) {
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-6f;
  // mem_every > 0: a 16-byte load per thread every mem_every iterations (4 MFMAs each), streamed from a 4 GB array: the
  // operand traffic of a GEMM tile (1 load per 32 MFMAs ~ 1.2 TB/s over the chip)
  size_t p = (static_cast<size_t>(blockIdx.x) * 4096 + threadIdx.x) & ((1ull << 28) - 1);
  for (int i = 0; i < iters; i++) {
    if (mem_every > 0 && (i % mem_every) == 0) { const float4 v = src[p]; a += v.x * 1e-9f; b += v.y * 1e-9f; p = (p + 256) & ((1ull << 28) - 1); }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
  }
  float s = 0;
  for (int k = 0; k < 16; k++) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
  if (s == 12345.f) out[0] = s;
}

__global__ __launch_bounds__(1024) void B(int mode, volatile int *stop, int *started, unsigned *mem) {
  extern __shared__ unsigned lds[];
  // ~100 live registers, so that the workgroup owns the CU's register file like a decoder lane
  unsigned r[96];
#pragma unroll
  for (int k = 0; k < 96; k++) r[k] = threadIdx.x * (k + 1);
  lds[threadIdx.x] = threadIdx.x;
  unsigned *mine = mem + static_cast<size_t>(blockIdx.x) * (16u << 20);
  if (threadIdx.x == 0) atomicAdd(started, 1);
  unsigned it = 0;
  while (true) {
    if (mode == 0) __builtin_amdgcn_s_sleep(64);
    else if (mode == 1) {
      for (int k = 0; k < 8; k++) { const unsigned p = ((it * 8 + k) * 1024 + threadIdx.x) & ((16u << 20) - 1); mine[p] = r[k] + it; r[k + 8] += mine[(p + 4096) & ((16u << 20) - 1)]; }
    } else {
      for (int k = 0; k < 4; k++) { const unsigned p = (threadIdx.x * 2654435761u + it * 40503u + k * 977u) & ((16u << 20) - 1); r[k] += atomicAdd(&mine[p], 1u); }
    }
#pragma unroll
    for (int k = 0; k < 96; k++) r[k] = r[k] * 1664525u + 1013904223u;
    it++;
    int s = 0;
    if (threadIdx.x == 0) s = *stop;
    s = __shfl(s, 0, 64);
    __shared__ int sstop;
    if (threadIdx.x == 0) sstop = s;
    __syncthreads();
    if (sstop) break;
    __syncthreads();
  }
  unsigned acc = 0;
#pragma unroll
  for (int k = 0; k < 96; k++) acc ^= r[k];
  if (acc == 0x12345u) mem[0] = acc + lds[(threadIdx.x + 1) & 1023];
}

int main() {
  int *h_stop, *started; unsigned *mem; float *out;
  CK(hipHostMalloc(reinterpret_cast<void **>(&h_stop), 4, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void **>(&started), 4, hipHostMallocDefault));
  CK(hipMalloc(&mem, 64ull * (64u << 20))); CK(hipMalloc(&out, 4));
  CK(hipMemset(mem, 0, 64ull * (64u << 20)));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(B), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int launches = 20, grid = 768 * 24, iters = 600;
  float4 *src; CK(hipMalloc(&src, (1ull << 28) * sizeof(float4)));
  CK(hipMemset(src, 0, (1ull << 28) * sizeof(float4)));
  for (int mem_every : {0, 8}) {
  auto run_a = [&]() {
    CK(hipEventRecord(e0, sa));
    for (int l = 0; l < launches; l++) A<<<grid, 256, 0, sa>>>(iters, out, src, mem_every);
    CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  run_a();
  const float alone = run_a();
  printf("A (%s) alone: %.2f ms for %d launches\n", mem_every ? "MFMA + streamed operand loads" : "register-only MFMA", alone, launches);
  for (int mode = 0; mode < 3; mode++)
    for (int n : {16, 64}) {
      *h_stop = 0; *started = 0;
      B<<<n, 1024, 96 * 1024, sb>>>(mode, h_stop, started, mem);
      while (*reinterpret_cast<volatile int *>(started) < n) usleep(100);
      const float t = run_a();
      *h_stop = 1;
      CK(hipStreamSynchronize(sb));
      printf("beside %2d occupiers, mode %d (%s): %.2f ms = %.2fx alone (proportional would be %.2fx)\n", n, mode,
             mode == 0 ? "spin" : mode == 1 ? "stream" : "atomics", t, t / alone, 256.0 / (256 - n));
    }
  }
  return 0;
}
