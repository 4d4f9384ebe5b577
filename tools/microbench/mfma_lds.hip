// The GEMM's inner loop without its global side: 4 waves (2 x 2), 128 x 128 tile, operands already in an LDS ring of three
// k-blocks; per k-block: [s_barrier], 8 ds_read_b128 per wave, 32 MFMAs per wave.  What do the LDS reads and the barrier
// cost against the register-only rate (mfma_peak)?   variants: barrier on/off, occupancy 1..3 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool BARRIER, int DMA>      // DMA: 0 none, 1 = the GEMM's global_load_lds refill of the ring (4 per wave and k-block) with its counted waits
__global__ __launch_bounds__(256, 3) void Loop(float *out, int nkb, const float *src, int src_mask, int rnd) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * (128 + 128) * 64];
  float *f = reinterpret_cast<float *>(smem);
  // operand data: `rnd` = 0: a smooth ramp (few bits toggle from one MFMA to the next), 1: pseudo-random normal-ish values
  for (int i = threadIdx.x; i < 3 * 256 * 16; i += 256) {
    unsigned h = (i + 1u) * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    f[i] = rnd ? (static_cast<float>(h & 0xFFFF) - 32768.0f) * (1.0f / 16384.0f) : 1e-3f * (i & 63);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)smem));
  const float *pg = src + ((blockIdx.x * 4096 + threadIdx.x * 4) & src_mask & ~3);
  auto issue = [&](int kb) {
    const unsigned stg = smem_lds + (kb % 3) * 256 * 64;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned m0v = __builtin_amdgcn_readfirstlane(stg + (wave + 4 * q) * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(pg) : "memory");
      pg = src + ((static_cast<unsigned>(pg - src) + 4096 + 16) & src_mask & ~3);
    }
  };
  if (DMA) { issue(0); issue(1); }   // (variants 3 and 4 start on stages filled the ordinary way)
  // variant 4: the GEMM's real gather: a piece = 16 rows x 64 bytes, rows 12 KB apart (a 3072-wide fp32 operand)
  const float *pr[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const size_t row = (static_cast<size_t>(blockIdx.x) * 256 + (wave + 4 * q) * 16 + (lane >> 2)) % 300000;
    pr[q] = src + row * 3072 + (lane & 3) * 4;
  }
  for (int kb = 0; kb < nkb; kb++) {
    if (DMA) {
      const int ahead = min(1, nkb - 1 - kb);
      if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (DMA == 1 && kb + 2 < nkb) issue(kb + 2);
    if (DMA == 4 && kb + 2 < nkb) {
      const unsigned stg = smem_lds + ((kb + 2) % 3) * 256 * 64;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(stg + (wave + 4 * q) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(pr[q]) : "memory");
        pr[q] += 16;                                   // the next 64 bytes of the same 16 rows
        if ((kb & 127) == 127) pr[q] -= 16 * 128;   // stay inside the rows (8 KB of each row is used)
      }
    }
    if (DMA == 3 && kb + 2 < nkb) {
      // one M0 per k-block: this wave's four 1 KB pieces are consecutive in LDS, the instruction offset moves both sides
      const unsigned m0v = __builtin_amdgcn_readfirstlane(smem_lds + ((kb + 2) % 3) * 256 * 64 + wave * 4096);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072" ::"s"(m0v), "v"(pg) : "memory");
      pg = src + ((static_cast<unsigned>(pg - src) + 4096 + 16) & src_mask & ~3);
    }
    const unsigned char *st = smem + (kb % 3) * 256 * 64;
    float4 a[2][2], b[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int m = wm * 64 + i * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < 2; tt++) a[i][tt] = *reinterpret_cast<const float4 *>(st + m * 64 + (((2 * tt + lk) ^ ((m >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = wn * 64 + j * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < 2; tt++) b[j][tt] = *reinterpret_cast<const float4 *>(st + 128 * 64 + n * 64 + (((2 * tt + lk) ^ ((n >> 2) & 3)) << 4));
    }
    if (DMA == 2 && kb + 2 < nkb) issue(kb + 2);
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const float av = q == 0 ? a[i][tt].x : q == 1 ? a[i][tt].y : q == 2 ? a[i][tt].z : a[i][tt].w;
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const float bv = q == 0 ? b[j][tt].x : q == 1 ? b[j][tt].y : q == 2 ? b[j][tt].z : b[j][tt].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
  }
  float s = 0.f;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool BARRIER, int DMA>
static void Run(const char *name, int cus, float *d_out, const float *d_src, int src_mask, int rnd = 0) {
  const int nkb = 40000;
  for (int occ = 1; occ <= 3; occ++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((Loop<BARRIER, DMA>), dim3(cus * occ), dim3(256), 0, 0, d_out, 100, d_src, src_mask, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((Loop<BARRIER, DMA>), dim3(cus * occ), dim3(256), 0, 0, d_out, nkb, d_src, src_mask, rnd);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 128 * 128 * 16 * static_cast<double>(nkb) * occ * cus;
    printf("%s, %d workgroup(s) per CU: %.1f ms, %.1f TFLOP/s\n", name, occ, ms, flops / (ms * 1e-3) / 1e12);
  }
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  float *d_out;
  (void)hipMalloc(&d_out, sizeof(float) * prop.multiProcessorCount * 3 * 256);
  float *d_small, *d_big;
  (void)hipMalloc(&d_small, ((1 << 20) + 64) * sizeof(float));            // 4 MB: lives in the L2s
  (void)hipMalloc(&d_big, ((1ull << 30) + 64) * sizeof(float));           // 4 GB: HBM
  (void)hipMemset(d_small, 0, (1 << 20) * sizeof(float));
  (void)hipMemset(d_big, 0, (1ull << 30) * sizeof(float));
  (void)hipDeviceSynchronize();
  Run<false, 0>("LDS reads, no barrier", prop.multiProcessorCount, d_out, d_small, (1 << 20) - 1);
  Run<true, 0>("LDS reads + s_barrier per k-block", prop.multiProcessorCount, d_out, d_small, (1 << 20) - 1);
  Run<true, 0>("LDS reads + s_barrier, RANDOM operand data", prop.multiProcessorCount, d_out, d_small, (1 << 20) - 1, 1);
  Run<true, 1>("the same + DMA refill from a 4 MB buffer", prop.multiProcessorCount, d_out, d_small, (1 << 20) - 1);
  Run<true, 1>("the same + DMA refill from a 4 GB buffer", prop.multiProcessorCount, d_out, d_big, (1 << 30) - 1);
  Run<true, 4>("DMA refill gathering 16 rows x 64 B per instruction, rows 12 KB apart (4 GB)", prop.multiProcessorCount, d_out, d_big, (1 << 30) - 1);
  Run<true, 2>("DMA issued after the fragment reads (4 GB)", prop.multiProcessorCount, d_out, d_big, (1 << 30) - 1);
  Run<true, 3>("one M0 per k-block, instruction offsets (4 GB)", prop.multiProcessorCount, d_out, d_big, (1 << 30) - 1);
  return 0;
}
