// What does one workgroup-wide step of the decoder's frame loop cost on a 1024-thread workgroup (one per CU, 256 CUs)?
//   mode 0: s_waitcnt lgkmcnt(0) + s_barrier                       (the kernel's LdsBarrier)
//   mode 1: a 64-bit min reduction: 6 shuffle steps, barrier, 16 LDS words, barrier, 16 reads   (CommitFrame2's kmin)
//   mode 2: mode 1 with two scattered 4-byte global stores per thread in front (the compaction's token records)
//   mode 3: mode 0 with one dependent global load (L2 hit) per thread in front
// build: hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip ; run: ./barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
__device__ inline void LdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__global__ __launch_bounds__(1024) void K(int mode, int iters, int *g, int stride, u64 *out) {
  __shared__ u64 red[16];
  __shared__ int sink;
  const int tid = threadIdx.x;
  int *mine = g + static_cast<size_t>(blockIdx.x) * (1 << 22);
  u64 acc = tid * 2654435761u;
  for (int it = 0; it < iters; it++) {
    if (mode == 2) {
      const int p = (tid * 37 + it * 1031) & ((1 << 20) - 1);
      mine[p] = it; mine[(1 << 20) + p] = tid;
    }
    if (mode == 3) acc += mine[(tid * stride + (it & 1023)) & ((1 << 20) - 1)];
    if (mode == 0 || mode == 3) { LdsBarrier(); continue; }
    u64 k = acc ^ (static_cast<u64>(it) << 20);
    for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(k, o, 64); k = t < k ? t : k; }
    LdsBarrier();
    if ((tid & 63) == 0) red[tid >> 6] = k;
    LdsBarrier();
    k = red[0];
    for (int i = 1; i < 16; i++) k = red[i] < k ? red[i] : k;
    acc += k;
  }
  if (acc == 1) sink = 1;
  if (tid == 0) out[blockIdx.x] = acc;
}
int main() {
  int *g; u64 *out;
  hipMalloc(&g, 256ull * (1 << 22) * 4); hipMalloc(&out, 256 * 8);
  hipMemset(g, 0, 256ull * (1 << 22) * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  for (int mode = 0; mode < 4; mode++) {
    K<<<256, 1024>>>(mode, 100, g, 1, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    K<<<256, 1024>>>(mode, iters, g, mode == 3 ? 1 : 0, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d: %.3f us per iteration\n", mode, 1000.0 * ms / iters);
  }
  return 0;
}
