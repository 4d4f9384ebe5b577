// What would a frame-loop barrier cost if a long utterance's frame were split over TWO workgroups (two CUs)?
// Pairs of 1024-thread workgroups meet at a two-party barrier through L2: LDS barrier, thread 0 does an agent-scope
// release add on the pair's counter and spins (acquire loads) until the partner has arrived, LDS barrier.  Reported:
// microseconds per meeting for partners on the SAME XCD (blockIdx b and b + 8: equal b % 8 under round-robin placement)
// and on DIFFERENT XCDs (b and b + 1), with an idle device (one pair) and with 128 pairs at once; and, beside it, the
// workgroup-local LdsBarrier it would replace.  The argument this feeds: DESIGN.md section 6 ("more than one CU per
// long utterance").
// build: hipcc --offload-arch=gfx950 -O3 -o xwg_barrier xwg_barrier.hip ; run: ./xwg_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ inline void LdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// pair p = (wg a, wg b); partner_stride: 8 (same XCD label) or 1 (neighbouring XCDs); mode 0: local barrier only
__global__ __launch_bounds__(1024) void K(int mode, int iters, int partner_stride, unsigned *counters, unsigned long long *cycles) {
  const int b = blockIdx.x, tid = threadIdx.x;
  // blocks are grouped in sets of 2 * partner_stride: block k of the first half pairs with block k of the second half
  const int set = b / (2 * partner_stride), in = b % (2 * partner_stride), k = in % partner_stride;
  unsigned *ctr = counters + 64 * (set * partner_stride + k);       // (a 256-byte line per pair)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 1; it <= iters; it++) {
    LdsBarrier();
    if (mode == 1 && tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 2u * it) __builtin_amdgcn_s_sleep(1);
    }
    LdsBarrier();
  }
  if (tid == 0) cycles[b] = __builtin_amdgcn_s_memrealtime() - t0;      // 100 MHz constant clock
}
int main() {
  const int iters = 20000;
  unsigned *d_ctr; unsigned long long *d_cyc;
  hipMalloc(&d_ctr, 64 * 4 * 256); hipMalloc(&d_cyc, 8 * 512);
  auto run = [&](int mode, int blocks, int stride, const char *what) {
    hipMemset(d_ctr, 0, 64 * 4 * 256);
    hipLaunchKernelGGL(K, dim3(blocks), dim3(1024), 0, 0, mode, iters, stride, d_ctr, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (auto v : c) mx = v > mx ? v : mx;
    printf("%-70s %7.3f us per meeting\n", what, mx * 10.0 / 1000.0 / iters);       // 100 MHz -> 10 ns per tick
  };
  run(0, 2, 1, "two LdsBarriers, no partner (the workgroup-local cost)");
  run(1, 16, 8, "two-party barrier through L2, partners on one XCD (b, b + 8), idle device");
  run(1, 2, 1, "two-party barrier through L2, partners on two XCDs (b, b + 1), idle device");
  run(1, 256, 8, "... partners on one XCD, 128 pairs at once");
  run(1, 256, 1, "... partners on two XCDs, 128 pairs at once");
  return 0;
}
