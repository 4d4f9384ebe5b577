// nnet.hip -- fused TDNN(-F) acoustic-model forward on gfx950 (fp32 MFMA).
//
// Replaces NnetComputer::Run over the compiled computation (nnet3/nnet-compute.cc:
// 508-540) for the component set of SURVEY 8(a9): every layer
//   y[t] = ((bn_scale * relu(sum_i W_i x[t+off_i] + W_iv ivec + b) + bn_offset
//            + bypass_scale * z[t]) + post_offset) * post_scale
// is ONE kernel: an fp32 GEMM on v_mfma_f32_32x32x2_f32 whose A rows are gathered
// through a per-(offset,row) index map (TdnnComponent's strided views,
// nnet-tdnn-component.cc:497-511, and Append/Offset descriptors) and whose epilogue
// carries bias / ReLU / BatchNorm(test) / bypass / -log-prior / acoustic-scale.
// The interpreter's ~6-10 launches per layer become one.
//
// Whole utterances are batched along M (no chunk-edge recomputation); each layer is
// evaluated only at the time indices its consumers need (frame-subsampling-factor 3
// is exploited from the first stride-3-compatible layer on, as the reference's
// compiler does).  Numerics: exact fp32 products, fp32 accumulate (k-ordered chain
// per output element => results do not depend on batch composition).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "meta_ring.h"

namespace kamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A; int ldA;          // producer activations
  const int *rowmap;                // [n_off][M] producer row per (offset, output row)
  int M, N, n_off, in_pad;          // K = n_off * in_pad (multiple of 16)
  const float *W;                   // [N_pad][K] zero padded
  const float *bias;                // [N] or NULL
  const float *ivbias; const int *row2utt;  // [n_utts][N], [M] or NULL
  int relu;
  const float *bn_scale, *bn_offset;
  const float *byp; int ld_byp; const int *bypmap; float bypass_scale;
  const float *post_offset; float post_scale;
  float *C; int ldC;
  int gx, gy;                       // > 0: 1-D launch, XCD-aware tile order (see the kernel)
  const float *zeros;               // >= 16 zero floats: the source of A rows that do not exist (rowmap < 0, m >= M)
  const float *zeros_n, *ones_n;    // >= N zeros / ones: the neutral operands of EpilogueWave
#ifdef KAMD_GEMM_LAB
  unsigned long long *stamps;       // tools/microbench/gemm_lab.hip: 32 words per workgroup (phase stamps, per-wave loop segments)
  int lab_loop;                     // also time the segments of every k-block (perturbs the loop)
  int lab_valu;                     // experiment: this many extra (useless) vector ALU instructions per k-block and wave
  int lab_aux;                      // experiment: cache policy of the operand DMAs (1 nt, 2 sc1, 3 sc0 sc1)
  int lab_prio;                     // experiment: wave priority 3 from the counted wait to the end of the DMA issue, 0 during the MFMAs
#endif
};
#ifdef KAMD_GEMM_LAB
#define KAMD_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define KAMD_STAMP_REAL(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KAMD_STAMP(i) do { } while (0)
#define KAMD_STAMP_REAL(i) do { } while (0)
#endif

// Epilogue of one 32x32 accumulator tile of one wavefront (round 2).  The MFMA C/D layout gives a lane 16 values of
// ONE column; stored from there every global access is a 4-byte one (64 scalar loads + 64 scalar stores per lane and
// workgroup tile with a bypass): on the K = 320 -> N = 1536 layers, whose main loop is only 20 k-blocks long, that tail
// was ~40 % of the kernel.  Here the tile goes through a per-wave LDS scratch ([32][36] floats, in the operand buffers
// the main loop no longer needs) and comes back row-major, so bias / BatchNorm / bypass / store are 16-byte accesses,
// eight lanes per 128-byte row segment.  Same arithmetic per element, same result bits.
#define EPI_LD 36
__device__ inline void EpilogueTile(const GemmArgs &p, const f32x16 &acc, int m_base, int n_base, float *scr /* [32][EPI_LD], this wave's */) {
  const int lane = threadIdx.x & 63, lr = lane & 31, lk = lane >> 5;
  const bool vec = (p.N & 3) == 0 && (p.ldC & 3) == 0 && (!p.byp || (p.ld_byp & 3) == 0);
  if (!vec) {   // unaligned shapes: the scalar path
    const int n = n_base + lr;
    if (n >= p.N) return;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const float bs = p.bn_scale ? p.bn_scale[n] : 1.f, bo = p.bn_scale ? p.bn_offset[n] : 0.f;
    const float po = p.post_offset ? p.post_offset[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (m >= p.M) continue;
      float v = acc[r] + bias;
      if (p.ivbias) v += p.ivbias[static_cast<size_t>(p.row2utt[m]) * p.N + n];
      if (p.relu) v = fmaxf(v, 0.f);
      if (p.bn_scale) v = v * bs + bo;
      if (p.byp) v += p.bypass_scale * p.byp[static_cast<size_t>(p.bypmap[m]) * p.ld_byp + n];
      if (p.post_offset) v += po;
      v *= p.post_scale;
      p.C[static_cast<size_t>(m) * p.ldC + n] = v;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_LD + lr] = acc[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the scratch is private to the wave: no barrier
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int c4 = (lane & 7) * 4, n = n_base + c4;
  if (n < p.N) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bias = p.bias ? *reinterpret_cast<const float4 *>(p.bias + n) : zero4;
    const float4 bs = p.bn_scale ? *reinterpret_cast<const float4 *>(p.bn_scale + n) : one4;
    const float4 bo = p.bn_scale ? *reinterpret_cast<const float4 *>(p.bn_offset + n) : zero4;
    const float4 po = p.post_offset ? *reinterpret_cast<const float4 *>(p.post_offset + n) : zero4;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int row = (lane >> 3) + 8 * pass, m = m_base + row;
      if (m >= p.M) continue;
      const float4 a = *reinterpret_cast<const float4 *>(scr + row * EPI_LD + c4);
      float v[4] = {a.x + bias.x, a.y + bias.y, a.z + bias.z, a.w + bias.w};
      if (p.ivbias) {
        const float4 iv = *reinterpret_cast<const float4 *>(p.ivbias + static_cast<size_t>(p.row2utt[m]) * p.N + n);
        v[0] += iv.x; v[1] += iv.y; v[2] += iv.z; v[3] += iv.w;
      }
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (p.bn_scale) { v[0] = v[0] * bs.x + bo.x; v[1] = v[1] * bs.y + bo.y; v[2] = v[2] * bs.z + bo.z; v[3] = v[3] * bs.w + bo.w; }
      if (p.byp) {
        const float4 z = *reinterpret_cast<const float4 *>(p.byp + static_cast<size_t>(p.bypmap[m]) * p.ld_byp + n);
        v[0] += p.bypass_scale * z.x; v[1] += p.bypass_scale * z.y; v[2] += p.bypass_scale * z.z; v[3] += p.bypass_scale * z.w;
      }
      if (p.post_offset) { v[0] += po.x; v[1] += po.y; v[2] += po.z; v[3] += po.w; }
      *reinterpret_cast<float4 *>(p.C + static_cast<size_t>(m) * p.ldC + n) =
          make_float4(v[0] * p.post_scale, v[1] * p.post_scale, v[2] * p.post_scale, v[3] * p.post_scale);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next tile reuses the scratch
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Epilogue, round 3.  EpilogueTile (above) walks a tile's four row passes one after the other, and in each pass the
// bypass operand is TWO dependent global loads (bypmap[m], then the row) behind `s_waitcnt vmcnt(0)`: sixteen passes per
// wave = 32 serialized L2 / HBM round trips per workgroup tile -- on the K = 320 -> N = 1536 layers (20 k-blocks of main
// loop) longer than the main loop's own MFMA time.  Here everything a tile needs from global memory -- bias, BatchNorm
// scale / offset, post offset (one float4 each) and the four bypass rows -- is requested up front as independent loads,
// two tiles ahead of its use (the bypass row numbers sit in LDS since the prologue), so a workgroup tile pays about one
// round trip instead of 32.  Arithmetic per element, and therefore every result bit, is EpilogueTile's.
struct EpiLoads { float4 bias, bs, bo, po, z[4]; };
// Straight-line code: a term a layer does not have is computed with neutral operands (bias / offsets from a vector of
// zeros, BatchNorm scale from a vector of ones, the bypass from row 0 of the zeros with scale 0, ReLU as max(v, -inf)),
// which leaves every bit of v as the conditional form would (x + 0, x * 1 + 0 and max(x, -inf) are exact; a -0 sum
// becomes +0).  With no branches between the requests and their uses hipcc counts the outstanding loads exactly
// (vmcnt(N)); behind any uniform branch it falls back to vmcnt(0) and the two-tile run-ahead is lost.
template <int TI, int TJ, int DEPTH, bool IVB>
__device__ inline void EpilogueWave(const GemmArgs &p, f32x16 (&acc)[TI][TJ], int m_wave, int n_wave, int row_wave /* first row of the wave inside the workgroup tile */,
                                    float *scr /* [32][EPI_LD], this wave's */, const int *bm /* LDS: bypass row of every tile row */) {
  constexpr int NT = TI * TJ;
  const int lane = threadIdx.x & 63, c4 = (lane & 7) * 4, r8 = lane >> 3;
  const float *bias = p.bias ? p.bias : p.zeros_n, *bsc = p.bn_scale ? p.bn_scale : p.ones_n, *bof = p.bn_scale ? p.bn_offset : p.zeros_n;
  const float *pof = p.post_offset ? p.post_offset : p.zeros_n, *byp = p.byp ? p.byp : p.zeros_n;
  const int ld_byp = p.byp ? p.ld_byp : 0;
  const float byps = p.byp ? p.bypass_scale : 0.f, floor_ = p.relu ? 0.f : -INFINITY, posts = p.post_scale;
  auto request = [&](int t, EpiLoads &L) {
    const int i = t / TJ, j = t % TJ;
    const int n = n_wave + j * 32 + c4, nc = min(n, p.N - 4);          // lanes beyond N read the last float4 and store nothing
    L.bias = *reinterpret_cast<const float4 *>(bias + nc);
    L.bs = *reinterpret_cast<const float4 *>(bsc + nc);
    L.bo = *reinterpret_cast<const float4 *>(bof + nc);
    L.po = *reinterpret_cast<const float4 *>(pof + nc);
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int row = row_wave + i * 32 + r8 + 8 * pass;                // rows beyond M and layers without a bypass: row 0
      L.z[pass] = *reinterpret_cast<const float4 *>(byp + static_cast<size_t>(bm[row]) * ld_byp + nc);
    }
  };
  EpiLoads L[DEPTH];
#pragma unroll
  for (int t = 0; t < DEPTH && t < NT; t++) request(t, L[t]);
  const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int i = t / TJ, j = t % TJ;
#pragma unroll
    for (int r = 0; r < 16; r++) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_LD + lr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the scratch is private to the wave: no barrier
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 a[4];
#pragma unroll
    for (int pass = 0; pass < 4; pass++) a[pass] = *reinterpret_cast<const float4 *>(scr + (r8 + 8 * pass) * EPI_LD + c4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next tile reuses the scratch
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const EpiLoads &E = L[t % DEPTH];
    const int n = n_wave + j * 32 + c4;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int m = m_wave + i * 32 + r8 + 8 * pass;
      float v[4] = {a[pass].x + E.bias.x, a[pass].y + E.bias.y, a[pass].z + E.bias.z, a[pass].w + E.bias.w};
      if (IVB) {                                                   // the first layer of a model with an i-vector input only
        const int mc = min(m, p.M - 1), nc = min(n, p.N - 4);
        const float4 iv = *reinterpret_cast<const float4 *>(p.ivbias + static_cast<size_t>(p.row2utt[mc]) * p.N + nc);
        v[0] += iv.x; v[1] += iv.y; v[2] += iv.z; v[3] += iv.w;
      }
      const float4 z = E.z[pass];
      v[0] = fmaxf(v[0], floor_); v[1] = fmaxf(v[1], floor_); v[2] = fmaxf(v[2], floor_); v[3] = fmaxf(v[3], floor_);
      v[0] = v[0] * E.bs.x + E.bo.x; v[1] = v[1] * E.bs.y + E.bo.y; v[2] = v[2] * E.bs.z + E.bo.z; v[3] = v[3] * E.bs.w + E.bo.w;
      v[0] += byps * z.x; v[1] += byps * z.y; v[2] += byps * z.z; v[3] += byps * z.w;
      v[0] += E.po.x; v[1] += E.po.y; v[2] += E.po.z; v[3] += E.po.w;
      if (m < p.M && n < p.N)
        *reinterpret_cast<float4 *>(p.C + static_cast<size_t>(m) * p.ldC + n) = make_float4(v[0] * posts, v[1] * posts, v[2] * posts, v[3] * posts);
    }
    if (t + DEPTH < NT) request(t + DEPTH, L[t % DEPTH]);
  }
}

// (256, 4): four workgroups per CU.  The main loop waits for its global prefetch and at a
// barrier once per k-block; what hides that is other workgroups on the same SIMDs, so the
// register budget is capped at 128 (measured: 48 -> 53 TFLOP/s at batch 64, 63 -> 70 at 512;
// BK = 32 halves the barriers but its LDS footprint drops the occupancy to 2: 43 / 51).
// BM x BN block tile, BK = 16, 256 threads = 4 waves laid out WM x WN, each wave owns
// TI x TJ MFMA tiles of 32x32.  LDS holds the tiles k-major ([k][m], +2 pad) so that
// fragment reads (lane -> consecutive m) and the transposing stores are conflict free.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 4) void TdnnGemmKernel(GemmArgs p) {
  constexpr int BK = 16;
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;
  constexpr int LDA = BM + 2, LDB = BN + 2;
  constexpr int OPER = 2 * BK * (LDA + LDB);
  static_assert(OPER >= 4 * 32 * EPI_LD, "the operand buffers double as the epilogue scratch of the four waves");
  __shared__ __attribute__((aligned(16))) float oper[OPER];
  float (*As)[BK][LDA] = reinterpret_cast<float (*)[BK][LDA]>(oper);
  float (*Bs)[BK][LDB] = reinterpret_cast<float (*)[BK][LDB]>(oper + 2 * BK * LDA);
  __shared__ int rm[KAMD_MAX_OFFSETS][BM];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  With a 2-D grid the
  // column tiles of one row block (which all stream the same A rows) would land on different
  // XCDs and every L2 would fetch the block once; the 1-D order below gives all column tiles
  // of a row block the same id mod 8 (speed only: any placement is correct).
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.gx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = xcd + 8 * (slot / p.gx); bx = slot % p.gx;
    if (by >= p.gy) return;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int K = p.n_off * p.in_pad;
  for (int i = t; i < p.n_off * BM; i += 256) {
    int o = i / BM, r = i % BM, m = m0 + r;
    rm[o][r] = (m < p.M) ? p.rowmap[static_cast<size_t>(o) * p.M + m] : -1;
  }
  __syncthreads();
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  constexpr int A_LOADS = BM * 4 / 256, B_LOADS = (BN * 4 + 255) / 256;  // float4 per thread
  float4 ra[A_LOADS], rb[B_LOADS];
  auto gload = [&](int kb) {
    const int k0 = kb * BK;
    const int off = k0 / p.in_pad, kk = k0 - off * p.in_pad;
#pragma unroll
    for (int q = 0; q < A_LOADS; q++) {
      int idx = t + q * 256, row = idx >> 2, kq = idx & 3;
      int src = rm[off][row];
      ra[q] = (src >= 0)
                  ? *reinterpret_cast<const float4 *>(p.A + static_cast<size_t>(src) * p.ldA + kk + kq * 4)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; q++) {
      int idx = t + q * 256, row = idx >> 2, kq = idx & 3;
      if (BN * 4 % 256 == 0 || row < BN)
        rb[q] = *reinterpret_cast<const float4 *>(p.W + static_cast<size_t>(n0 + row) * K + k0 + kq * 4);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < A_LOADS; q++) {
      int idx = t + q * 256, row = idx >> 2, kq = idx & 3;
      As[buf][kq * 4 + 0][row] = ra[q].x; As[buf][kq * 4 + 1][row] = ra[q].y;
      As[buf][kq * 4 + 2][row] = ra[q].z; As[buf][kq * 4 + 3][row] = ra[q].w;
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; q++) {
      int idx = t + q * 256, row = idx >> 2, kq = idx & 3;
      if (BN * 4 % 256 == 0 || row < BN) {
        Bs[buf][kq * 4 + 0][row] = rb[q].x; Bs[buf][kq * 4 + 1][row] = rb[q].y;
        Bs[buf][kq * 4 + 2][row] = rb[q].z; Bs[buf][kq * 4 + 3][row] = rb[q].w;
      }
    }
  };
  const int nkb = K / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int lr = lane & 31, lk = lane >> 5;
  for (int kb = 0; kb < nkb; kb++) {
    const int buf = kb & 1;
    if (kb + 1 < nkb) gload(kb + 1);   // global prefetch overlaps the MFMAs below
#pragma unroll
    for (int ks = 0; ks < BK / 2; ks++) {
      float a[TI], b[TJ];
#pragma unroll
      for (int i = 0; i < TI; i++) a[i] = As[buf][2 * ks + lk][wm * (BM / WM) + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TJ; j++) b[j] = Bs[buf][2 * ks + lk][wn * (BN / WN) + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nkb) sstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue (the loop ended with a barrier: nobody reads the operand buffers any more)
  asm volatile("" ::: "memory");      // keep the epilogue's vector loads out of the main loop's register budget
  float *scr = oper + wave * 32 * EPI_LD;
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
      EpilogueTile(p, acc[i][j], m0 + wm * (BM / WM) + i * 32, n0 + wn * (BN / WN) + j * 32, scr);
}

// Second-generation main loop (round 2): the same tile, the same exact-fp32 MFMA, but the operands go HBM/L2 -> LDS
// directly (global_load_lds_dwordx4, no staging registers) through a NST-stage ring, two k-blocks ahead of the MFMAs,
// with counted vmcnt waits and raw barriers.  The first generation (above) prefetches ONE 16-deep k-block through
// registers: ~1 us of MFMA work to hide a load that takes 2-3 us when every CU streams, and every wave of the workgroup
// then sits at the barrier (MfmaUtil 53 % at M = 4e5).
//   * LDS image of a stage: rows of 64 bytes ([m][16 k], A tile then B tile), as the DMA writes them (lane-linear: one
//     wave instruction = 16 rows x 4 sixteen-byte chunks).  Bank conflicts are avoided on the SOURCE side: the chunk
//     that lands in slot p of row m is k-chunk p ^ ((m >> 2) & 3), so the four rows that share a 256-byte bank row
//     and the rows 4, 8, 12 apart hold a given k-chunk in different slots (ds_read_b128 of 16 lanes: conflict free).
//   * one ds_read_b128 gives a lane 4 consecutive k of its row; lane half lk takes k-chunk 2t + lk, so MFMA step
//     (t, q) multiplies k = 8t + q (lanes 0-31) and k = 8t + 4 + q (lanes 32-63): every k once, 2 reads per operand
//     tile and k-block instead of 8.  The summation order inside a k-block differs from the first generation's
//     (a fixed permutation, the same for every batch composition): results stay exact-fp32 k-chains, bit-equal between
//     batch / streaming / chunked evaluation, but not bit-equal to generation 1.
//   * rows that do not exist (rowmap < 0: clamped context is explicit in the map, so this only pads M) read p.zeros.
// BK = 32 (round 3): rows of 128 bytes, i.e. whole cache lines per request (with BK = 16 every 128-byte line of an operand
// row is requested twice, half a line at a time, and a workgroup stalls ~570 cycles per DMA instruction in the issue:
// gemm_lab's per-k-block segments); 8 rows per DMA instruction, 8 sixteen-byte chunks per row, chunk p of row m holds
// k-chunk p ^ ((m >> 1) & 7) (two rows per 256-byte bank row: conflict free for ds_read_b128 as before).
// s_waitcnt vmcnt(ahead * LOADS): all but the newest `ahead` k-blocks of this wave's DMAs have landed (ahead <= MAXA)
template <int LOADS, int MAXA>
__device__ inline void WaitAhead(int ahead) {
  static_assert(MAXA * LOADS <= 63, "vmcnt is a 6-bit counter");
  if constexpr (MAXA > 0) {
    if (ahead >= MAXA) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXA * LOADS) : "memory"); return; }
    WaitAhead<LOADS, MAXA - 1>(ahead);
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int BM, int BN, int WM, int WN, int NST, int EPI = 2, int BK = 16, int OCC = 3>       // EPI 1: EpilogueTile (round 2), 2: EpilogueWave, 3: EpilogueWave + per-utterance i-vector bias
__global__ __launch_bounds__(256, OCC) void TdnnGemmDmaKernel(GemmArgs p) {
  static_assert(BK == 16 || BK == 32, "k-block depth");
  constexpr int ROWB = BK * 4, CH = ROWB / 16, RPI = 1024 / ROWB;    // bytes per operand row and stage, 16-byte chunks per row, rows per DMA instruction
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM / RPI, B_INST = BN / RPI;                // wave instructions per stage (1 KB each)
  constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;    // per wave (a clamped duplicate pads the count)
  constexpr int LOADS = A_PW + B_PW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES + (KAMD_MAX_OFFSETS + 1) * BM * 4];
  int *rm = reinterpret_cast<int *>(smem + NST * ST_BYTES);          // [n_off][BM]
  int *bm = rm + KAMD_MAX_OFFSETS * BM;                              // [BM] bypass rows (EPI 2)
  KAMD_STAMP(0); KAMD_STAMP_REAL(6);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.gx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = xcd + 8 * (slot / p.gx); bx = slot % p.gx;
    if (by >= p.gy) return;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int K = p.n_off * p.in_pad;
  for (int i = t; i < p.n_off * BM; i += 256) {
    const int o = i / BM, r = i % BM, m = m0 + r;
    rm[o * BM + r] = (m < p.M) ? p.rowmap[static_cast<size_t>(o) * p.M + m] : -1;
  }
  if (EPI >= 2)
    for (int r = t; r < BM; r += 256) bm[r] = (p.byp && m0 + r < p.M) ? p.bypmap[m0 + r] : 0;
  __syncthreads();
  KAMD_STAMP(1);
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  // The DMA is issued through inline asm: hipcc (ROCm 7.2) tracks a __builtin_amdgcn_global_load_lds as a pending LDS
  // write and puts s_waitcnt vmcnt(0) in front of the next ds_read of the same array, which would drain the two
  // k-blocks in flight on every iteration.  Invisible to that pass, the loads are ordered by the counted waits below.
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)smem));
  auto dma16 = [&](const float *g, unsigned lds_addr) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "memory");
  };
  const int lrow = lane / CH, lslot = lane % CH;
  auto swz = [](int row) { return BK == 16 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
  // Per-lane source pointers live in registers and advance by one k-block (64 bytes) per issue; only when the k index
  // crosses into the next time-offset slice of the TDNN operand are the A pointers rebuilt from the row map (the 64-bit
  // row * ldA arithmetic of every issue was ~60 VALU instructions per k-block beside 32 MFMAs).
  const float *pa[A_PW], *pb[B_PW];
  int ainc[A_PW];
  unsigned a_lds[A_PW], b_lds[B_PW];
#pragma unroll
  for (int q = 0; q < B_PW; q++) {
    int J = wave + 4 * q; if (J > B_INST - 1) J = B_INST - 1;
    const int row = RPI * J + lrow, c = lslot ^ swz(row);
    pb[q] = p.W + static_cast<size_t>(n0 + row) * K + 4 * c;
    b_lds[q] = A_BYTES + J * 1024;
  }
#pragma unroll
  for (int q = 0; q < A_PW; q++) { int I = wave + 4 * q; if (I > A_INST - 1) I = A_INST - 1; a_lds[q] = I * 1024; pa[q] = p.zeros; ainc[q] = 0; }
  const int kb_per_off = p.in_pad / BK;
  auto issue = [&](int kb) {
    if (kb % kb_per_off == 0) {                      // uniform: a new time-offset slice starts
      const int off = kb / kb_per_off;
#pragma unroll
      for (int q = 0; q < A_PW; q++) {
        const int row = a_lds[q] / ROWB + lrow, c = lslot ^ swz(row);
        const int src = rm[off * BM + row];
        pa[q] = src >= 0 ? p.A + static_cast<size_t>(src) * p.ldA + 4 * c : p.zeros + 4 * c;
        ainc[q] = src >= 0 ? BK : 0;
      }
    }
    const unsigned st = smem_lds + (kb % NST) * ST_BYTES;
#pragma unroll
    for (int q = 0; q < A_PW; q++) { dma16(pa[q], st + a_lds[q]); pa[q] += ainc[q]; }
#pragma unroll
    for (int q = 0; q < B_PW; q++) { dma16(pb[q], st + b_lds[q]); pb[q] += BK; }
  };
  const int nkb = K / BK;
  constexpr int DIST = NST - 1;          // k-blocks in flight ahead of the MFMAs
#pragma unroll
  for (int pkb = 0; pkb < DIST; pkb++) if (pkb < nkb) issue(pkb);
  KAMD_STAMP(2);
  const int lr = lane & 31, lk = lane >> 5;
#ifdef KAMD_GEMM_LAB
  int lab_dummy = 0;
  unsigned long long lab_t[4] = {0, 0, 0, 0}, lab_prev = 0;
#define KAMD_LAB_SEG(i) do { if (p.stamps && p.lab_loop) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); lab_t[i] += now_ - lab_prev; lab_prev = now_; } } while (0)
  if (p.stamps && p.lab_loop) lab_prev = __builtin_amdgcn_s_memtime();
#else
#define KAMD_LAB_SEG(i) do { } while (0)
#endif
  for (int kb = 0; kb < nkb; kb++) {
    // k-block kb has landed once all but the newest (DIST - 1) * LOADS of this wave's DMAs are done; then everybody's have
    const int ahead = min(DIST - 1, nkb - 1 - kb);
#ifdef KAMD_GEMM_LAB
    if (p.lab_prio) __builtin_amdgcn_s_setprio(3);
#endif
    WaitAhead<LOADS, DIST - 1>(ahead);
    KAMD_LAB_SEG(0);
    __builtin_amdgcn_s_barrier();
    KAMD_LAB_SEG(1);
#ifdef KAMD_GEMM_LAB
    if (kb == 0) KAMD_STAMP(3);
#endif
    if (kb + DIST < nkb) issue(kb + DIST);     // into the stage everybody finished reading before this barrier
#ifdef KAMD_GEMM_LAB
    if (p.lab_prio) __builtin_amdgcn_s_setprio(0);
    for (int i = 0; i < p.lab_valu; i++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(lab_dummy));
#endif
    KAMD_LAB_SEG(2);
    const unsigned char *st = smem + (kb % NST) * ST_BYTES;
    constexpr int NTT = BK / 8;
    float4 a[TI][NTT], b[TJ][NTT];
#pragma unroll
    for (int i = 0; i < TI; i++) {
      const int m = wm * (BM / WM) + i * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < NTT; tt++)
        a[i][tt] = *reinterpret_cast<const float4 *>(st + m * ROWB + (((2 * tt + lk) ^ swz(m)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      const int n = wn * (BN / WN) + j * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < NTT; tt++)
        b[j][tt] = *reinterpret_cast<const float4 *>(st + A_BYTES + n * ROWB + (((2 * tt + lk) ^ swz(n)) << 4));
    }
#pragma unroll
    for (int tt = 0; tt < NTT; tt++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 0; i < TI; i++) {
          const float av = q == 0 ? a[i][tt].x : q == 1 ? a[i][tt].y : q == 2 ? a[i][tt].z : a[i][tt].w;
#pragma unroll
          for (int j = 0; j < TJ; j++) {
            const float bv = q == 0 ? b[j][tt].x : q == 1 ? b[j][tt].y : q == 2 ? b[j][tt].z : b[j][tt].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    KAMD_LAB_SEG(3);
  }
#ifdef KAMD_GEMM_LAB
  if (p.stamps && p.lab_loop && lane == 0)
    for (int i = 0; i < 4; i++) p.stamps[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 32 + 8 + wave * 4 + i] = lab_t[i];
#endif
  KAMD_STAMP(4);
  __builtin_amdgcn_s_barrier();            // every wave has read the last stage: the ring becomes epilogue scratch
  asm volatile("" ::: "memory");
  float *scr = reinterpret_cast<float *>(smem) + wave * 32 * EPI_LD;
  if (EPI >= 2) {        // the host picks EPI 1 for shapes that are not float4-aligned
    EpilogueWave<TI, TJ, ((TI * TJ * 16 <= 64 && EPI != 3) ? 2 : 1), EPI == 3>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), wm * (BM / WM), scr, bm);
  } else {
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
        EpilogueTile(p, acc[i][j], m0 + wm * (BM / WM) + i * 32, n0 + wn * (BN / WN) + j * 32, scr);
  }
  KAMD_STAMP(5); KAMD_STAMP_REAL(7);
}

// Fourth generation (round 3): no vector ALU instructions in the main loop, few in the epilogue.
// What gemm_lab measured: on gfx950 the fp32 MFMA runs at the vector FMA rate (64 FLOP / clk / SIMD), and every vector
// ALU instruction a co-resident wave issues takes ~10 cycles away from it -- 16 / 32 / 64 useless v_add per k-block and
// wave cost the K = 3072 probe 4 / 12 / 26 % -- while the kernels above spend ~30 (per-lane 64-bit source pointers,
// v_readfirstlane for M0, stage address arithmetic) per k-block and ~130 per epilogue tile (every term of the layer
// formula as separate multiplies and adds on neutral operands).  Here:
//   * DMA sources stay per-lane 64-bit pointers, advanced by ONE v_lshl_add_u64 per piece and k-block (4 per wave): the
//     only vector ALU instructions of a k-block;
//   * M0 and the ring stage are scalar and compile-time (the k loop is unrolled by the three stages), so the fragment
//     reads are ds_read_b128 with immediate offsets from four per-lane base addresses;
//   * the epilogue is compiled per layer form (EF: bias / ReLU / BatchNorm / bypass / post offset / post scale) -- no
//     neutral operands -- with packed fp32 instructions and fused multiply-adds for the BatchNorm and bypass terms
//     (one rounding instead of two; within the 1e-4 of the nnet parity tests, and identical for every batch
//     composition).
// Same LDS image and k order as the second generation: the GEMM sums are bit-equal to it.
enum { EF_BIAS = 1, EF_RELU = 2, EF_BN = 4, EF_BYP = 8, EF_PO = 16, EF_SCALE = 32 };
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct EpiLoadsS { float4 bias, bs, bo, po, z[4]; };
template <int TI, int TJ, int EF, bool IVB>
__device__ inline void EpilogueSpec(const GemmArgs &p, f32x16 (&acc)[TI][TJ], int m_wave, int n_wave, int row_wave, float *scr, const int *bm) {
  constexpr int NT = TI * TJ;
  const int lane = threadIdx.x & 63, c4 = (lane & 7) * 4, r8 = lane >> 3;
  const int lr = lane & 31, lk = lane >> 5;
  auto request = [&](int t, EpiLoadsS &L) {
    const int i = t / TJ, j = t % TJ;
    const int nc = min(n_wave + j * 32 + c4, p.N - 4);
    if (EF & EF_BIAS) L.bias = *reinterpret_cast<const float4 *>(p.bias + nc);
    if (EF & EF_BN) { L.bs = *reinterpret_cast<const float4 *>(p.bn_scale + nc); L.bo = *reinterpret_cast<const float4 *>(p.bn_offset + nc); }
    if (EF & EF_PO) L.po = *reinterpret_cast<const float4 *>(p.post_offset + nc);
    if (EF & EF_BYP) {
#pragma unroll
      for (int pass = 0; pass < 4; pass++)
        L.z[pass] = *reinterpret_cast<const float4 *>(p.byp + static_cast<size_t>(bm[row_wave + i * 32 + r8 + 8 * pass]) * p.ld_byp + nc);
    }
  };
  EpiLoadsS L;
  request(0, L);
  const f32x2 byps = {p.bypass_scale, p.bypass_scale}, posts = {p.post_scale, p.post_scale};
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int i = t / TJ, j = t % TJ;
#pragma unroll
    for (int r = 0; r < 16; r++) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_LD + lr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 a[4];
#pragma unroll
    for (int pass = 0; pass < 4; pass++) a[pass] = *reinterpret_cast<const float4 *>(scr + (r8 + 8 * pass) * EPI_LD + c4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = n_wave + j * 32 + c4;
    float4 out[4];
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      f32x2 v0 = {a[pass].x, a[pass].y}, v1 = {a[pass].z, a[pass].w};
      if (EF & EF_BIAS) { v0 += f32x2{L.bias.x, L.bias.y}; v1 += f32x2{L.bias.z, L.bias.w}; }
      if (IVB) {
        const int mc = min(m_wave + i * 32 + r8 + 8 * pass, p.M - 1), nc = min(n, p.N - 4);
        const float4 iv = *reinterpret_cast<const float4 *>(p.ivbias + static_cast<size_t>(p.row2utt[mc]) * p.N + nc);
        v0 += f32x2{iv.x, iv.y}; v1 += f32x2{iv.z, iv.w};
      }
      if (EF & EF_RELU) { v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); }
      if (EF & EF_BN) {
        v0 = __builtin_elementwise_fma(v0, f32x2{L.bs.x, L.bs.y}, f32x2{L.bo.x, L.bo.y});
        v1 = __builtin_elementwise_fma(v1, f32x2{L.bs.z, L.bs.w}, f32x2{L.bo.z, L.bo.w});
      }
      if (EF & EF_BYP) {
        v0 = __builtin_elementwise_fma(byps, f32x2{L.z[pass].x, L.z[pass].y}, v0);
        v1 = __builtin_elementwise_fma(byps, f32x2{L.z[pass].z, L.z[pass].w}, v1);
      }
      if (EF & EF_PO) { v0 += f32x2{L.po.x, L.po.y}; v1 += f32x2{L.po.z, L.po.w}; }
      if (EF & EF_SCALE) { v0 *= posts; v1 *= posts; }
      out[pass] = make_float4(v0.x, v0.y, v1.x, v1.y);
    }
    if (t + 1 < NT) request(t + 1, L);      // (the values of tile t are consumed: the same registers take tile t + 1's)
    float *crow = p.C + static_cast<size_t>(m_wave + i * 32 + r8) * p.ldC + n;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int m = m_wave + i * 32 + r8 + 8 * pass;
      if (m < p.M && n < p.N) *reinterpret_cast<float4 *>(crow + static_cast<size_t>(8 * pass) * p.ldC) = out[pass];
    }
  }
}

template <int BM, int BN, int WM, int WN, int EF, bool IVB, int NST = 3, int OCC = 3>
__global__ __launch_bounds__(256, OCC) void TdnnGemmSaKernel(GemmArgs p) {
  constexpr int BK = 16, DIST = NST - 1;
  static_assert(NST == 2 || NST == 3, "ring depth");
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM / 16, B_INST = BN / 16;
  constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;
  constexpr int LOADS = A_PW + B_PW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES + (KAMD_MAX_OFFSETS + 1) * BM * 4];
  int *rm = reinterpret_cast<int *>(smem + NST * ST_BYTES);          // [n_off][BM]
  int *bm = rm + KAMD_MAX_OFFSETS * BM;                              // [BM]
  KAMD_STAMP(0); KAMD_STAMP_REAL(6);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.gx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = xcd + 8 * (slot / p.gx); bx = slot % p.gx;
    if (by >= p.gy) return;
  }
  bx = __builtin_amdgcn_readfirstlane(bx); by = __builtin_amdgcn_readfirstlane(by);    // (the division runs on the vector ALU: without this every base below lives in VGPRs)
  const int m0 = by * BM, n0 = bx * BN;
  const int K = p.n_off * p.in_pad, nkb = K / BK, kb_per_off = p.in_pad / BK;
  for (int i = t; i < p.n_off * BM; i += 256) {
    const int o = i / BM, r = i % BM;
    rm[o * BM + r] = p.rowmap[static_cast<size_t>(o) * p.M + min(m0 + r, p.M - 1)];
  }
  if (EF & EF_BYP)
    for (int r = t; r < BM; r += 256) bm[r] = p.bypmap[min(m0 + r, p.M - 1)];
  __syncthreads();
  KAMD_STAMP(1);
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned smem_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)smem)));
  const int lrow = lane >> 2, lslot = lane & 3;
  // per-lane source pointers (64-bit: one v_lshl_add_u64 per piece and k-block is all the vector ALU work of the loop;
  // a scalar base + 32-bit per-lane offset would need none, but hipcc keeps loop-carried 64-bit scalars in VGPRs once
  // the SGPR file is under pressure, and this kernel's argument block alone fills half of it)
  const float *pa[A_PW], *pb[B_PW];
  int a_piece[A_PW], b_piece[B_PW];            // uniform
#pragma unroll
  for (int q = 0; q < B_PW; q++) {
    b_piece[q] = min(wave + 4 * q, B_INST - 1);
    const int row = 16 * b_piece[q] + lrow, c = lslot ^ ((row >> 2) & 3);
    pb[q] = p.W + static_cast<size_t>(n0 + row) * K + 4 * c;
  }
#pragma unroll
  for (int q = 0; q < A_PW; q++) { a_piece[q] = min(wave + 4 * q, A_INST - 1); pa[q] = p.zeros; }
  int to_slice = 0, slice = 0;                 // k-blocks until the next time-offset slice starts (uniform)
  auto issue = [&](unsigned stage_bytes) {
    if (to_slice == 0) {
#pragma unroll
      for (int q = 0; q < A_PW; q++) {
        const int row = 16 * a_piece[q] + lrow, c = lslot ^ ((row >> 2) & 3);
        pa[q] = p.A + static_cast<size_t>(rm[slice * BM + row]) * p.ldA + 4 * c;
      }
      to_slice = kb_per_off; slice++;
    }
    to_slice--;
#pragma unroll
    for (int q = 0; q < A_PW; q++) {
      const unsigned m0v = smem_lds + stage_bytes + a_piece[q] * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(pa[q]) : "memory");
      pa[q] += BK;
    }
#pragma unroll
    for (int q = 0; q < B_PW; q++) {
      const unsigned m0v = smem_lds + stage_bytes + A_BYTES + b_piece[q] * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(pb[q]) : "memory");
      pb[q] += BK;
    }
  };
  if (nkb > 0) issue(0);
  if (DIST > 1 && nkb > 1) issue(ST_BYTES);
  KAMD_STAMP(2);
  const int lr = lane & 31, lk = lane >> 5;
  // per-lane fragment addresses inside a stage: rows lr (+ 32 i, same swizzle), k-chunks lk and 2 + lk
  const unsigned char *fa[2], *fb[2];
#pragma unroll
  for (int tt = 0; tt < 2; tt++) {
    const int m = wm * (BM / WM) + lr, n = wn * (BN / WN) + lr;
    fa[tt] = smem + m * 64 + (((2 * tt + lk) ^ ((m >> 2) & 3)) << 4);
    fb[tt] = smem + A_BYTES + n * 64 + (((2 * tt + lk) ^ ((n >> 2) & 3)) << 4);
  }
  int kb = 0;
  auto step = [&](auto stage_c) {
    constexpr int S = decltype(stage_c)::value;
    const int ahead = min(DIST - 1, nkb - 1 - kb);
    if (DIST > 1 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef KAMD_GEMM_LAB
    if (kb == 0) KAMD_STAMP(3);
#endif
    if (kb + DIST < nkb) issue(((S + DIST) % NST) * ST_BYTES);
    float4 a[TI][2], b[TJ][2];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int tt = 0; tt < 2; tt++) a[i][tt] = *reinterpret_cast<const float4 *>(fa[tt] + S * ST_BYTES + i * 32 * 64);
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int tt = 0; tt < 2; tt++) b[j][tt] = *reinterpret_cast<const float4 *>(fb[tt] + S * ST_BYTES + j * 32 * 64);
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 0; i < TI; i++) {
          const float av = q == 0 ? a[i][tt].x : q == 1 ? a[i][tt].y : q == 2 ? a[i][tt].z : a[i][tt].w;
#pragma unroll
          for (int j = 0; j < TJ; j++) {
            const float bv = q == 0 ? b[j][tt].x : q == 1 ? b[j][tt].y : q == 2 ? b[j][tt].z : b[j][tt].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    kb++;
  };
  while (kb + NST <= nkb) {
    step(std::integral_constant<int, 0>());
    step(std::integral_constant<int, 1>());
    if (NST == 3) step(std::integral_constant<int, 2>());
  }
  if (kb < nkb) step(std::integral_constant<int, 0>());
  if (NST == 3 && kb < nkb) step(std::integral_constant<int, 1>());
  KAMD_STAMP(4);
  __builtin_amdgcn_s_barrier();            // every wave has read the last stage: the ring becomes epilogue scratch
  asm volatile("" ::: "memory");
  float *scr = reinterpret_cast<float *>(smem) + wave * 32 * EPI_LD;
  EpilogueSpec<TI, TJ, EF, IVB>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), wm * (BM / WM), scr, bm);
  KAMD_STAMP(5); KAMD_STAMP_REAL(7);
}

// Loader-wave variant (round 3): gemm_lab's per-k-block stamps show a wave of the kernel above spending as long in the
// ISSUE of its four DMA instructions (~2300 cycles: the vector-memory queue pushes back while the compute unit's other
// workgroups stream too) as in its 32 MFMAs, and a wave stuck in the issue multiplies nothing.  Here a fifth wave does
// nothing but request operands -- all (BM + BN) / 16 pieces of a k-block -- and takes the back-pressure; the four MFMA
// waves execute barrier, fragment reads, MFMAs.  Same ring, same barrier per k-block (the loader waits for its own
// DMAs with a counted vmcnt in front of it), same LDS image and summation order: results are bit-equal.
template <int BM, int BN, int WM, int WN, int NST, int OCC, bool IVB>
__global__ __launch_bounds__(320, OCC) void TdnnGemmLoaderKernel(GemmArgs p) {
  constexpr int BK = 16;
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM / 16, B_INST = BN / 16, LOADS = A_INST + B_INST;
  constexpr int DIST = NST - 1;
  static_assert((DIST - 1) * LOADS <= 63 || DIST == 1, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES + (KAMD_MAX_OFFSETS + 1) * BM * 4];
  int *rm = reinterpret_cast<int *>(smem + NST * ST_BYTES);
  int *bm = rm + KAMD_MAX_OFFSETS * BM;
  KAMD_STAMP(0); KAMD_STAMP_REAL(6);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.gx > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    by = xcd + 8 * (slot / p.gx); bx = slot % p.gx;
    if (by >= p.gy) return;
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int K = p.n_off * p.in_pad, nkb = K / BK;
  for (int i = t; i < p.n_off * BM; i += 320) {
    const int o = i / BM, r = i % BM;
    rm[o * BM + r] = p.rowmap[static_cast<size_t>(o) * p.M + min(m0 + r, p.M - 1)];
  }
  for (int r = t; r < BM; r += 320) bm[r] = p.byp ? p.bypmap[min(m0 + r, p.M - 1)] : 0;
  __syncthreads();
  KAMD_STAMP(1);
  if (wave == 4) {
    // ------------------------------------------------------------------ the loader
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)smem));
    const int lrow = lane >> 2, lslot = lane & 3;
    const float *pa[A_INST], *pb[B_INST];
#pragma unroll
    for (int J = 0; J < B_INST; J++) {
      const int row = 16 * J + lrow, c = lslot ^ ((row >> 2) & 3);
      pb[J] = p.W + static_cast<size_t>(n0 + row) * K + 4 * c;
    }
#pragma unroll
    for (int I = 0; I < A_INST; I++) pa[I] = p.zeros;
    const int kb_per_off = p.in_pad / BK;
    auto issue = [&](int kb) {
      if (kb % kb_per_off == 0) {
        const int off = kb / kb_per_off;
#pragma unroll
        for (int I = 0; I < A_INST; I++) {
          const int row = 16 * I + lrow, c = lslot ^ ((row >> 2) & 3);
          pa[I] = p.A + static_cast<size_t>(rm[off * BM + row]) * p.ldA + 4 * c;
        }
      }
      const unsigned st = smem_lds + (kb % NST) * ST_BYTES;
#pragma unroll
      for (int I = 0; I < A_INST; I++) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(st + I * 1024), "v"(pa[I]) : "memory");
        pa[I] += BK;
      }
#pragma unroll
      for (int J = 0; J < B_INST; J++) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(st + A_BYTES + J * 1024), "v"(pb[J]) : "memory");
        pb[J] += BK;
      }
    };
#pragma unroll
    for (int pkb = 0; pkb < DIST; pkb++) if (pkb < nkb) issue(pkb);
    for (int kb = 0; kb < nkb; kb++) {
      const int ahead = min(DIST - 1, nkb - 1 - kb);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS > 63 ? 63 : 2 * LOADS) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kb + DIST < nkb) issue(kb + DIST);
    }
    __builtin_amdgcn_s_barrier();
    return;
  }
  // -------------------------------------------------------------------- the four MFMA waves
  const int wm = wave / WN, wn = wave % WN;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const int lr = lane & 31, lk = lane >> 5;
  KAMD_STAMP(2);
  for (int kb = 0; kb < nkb; kb++) {
    __builtin_amdgcn_s_barrier();
#ifdef KAMD_GEMM_LAB
    if (kb == 0) KAMD_STAMP(3);
#endif
    const unsigned char *st = smem + (kb % NST) * ST_BYTES;
    float4 a[TI][2], b[TJ][2];
#pragma unroll
    for (int i = 0; i < TI; i++) {
      const int m = wm * (BM / WM) + i * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < 2; tt++)
        a[i][tt] = *reinterpret_cast<const float4 *>(st + m * 64 + (((2 * tt + lk) ^ ((m >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      const int n = wn * (BN / WN) + j * 32 + lr;
#pragma unroll
      for (int tt = 0; tt < 2; tt++)
        b[j][tt] = *reinterpret_cast<const float4 *>(st + A_BYTES + n * 64 + (((2 * tt + lk) ^ ((n >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 0; i < TI; i++) {
          const float av = q == 0 ? a[i][tt].x : q == 1 ? a[i][tt].y : q == 2 ? a[i][tt].z : a[i][tt].w;
#pragma unroll
          for (int j = 0; j < TJ; j++) {
            const float bv = q == 0 ? b[j][tt].x : q == 1 ? b[j][tt].y : q == 2 ? b[j][tt].z : b[j][tt].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
  }
  KAMD_STAMP(4);
  __builtin_amdgcn_s_barrier();            // every wave has read the last stage: the ring becomes epilogue scratch
  asm volatile("" ::: "memory");
  float *scr = reinterpret_cast<float *>(smem) + wave * 32 * EPI_LD;
  EpilogueWave<TI, TJ, 1, IVB>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), wm * (BM / WM), scr, bm);
  KAMD_STAMP(5); KAMD_STAMP_REAL(7);
}

// Third generation (round 3): the same tile, ring and MFMA schedule, but a workgroup is PERSISTENT: it walks the tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... and its DMA stream never stops at a tile boundary -- the first k-blocks of the
// next tile are requested while the last k-blocks of this one are multiplied, and land while the epilogue runs.  In the
// one-tile kernel above a workgroup spends ~26 k cycles (of ~150 k on the K = 320 layers) between its launch and its
// first MFMA: row map global -> LDS, barrier, source pointers, ring fill, DMA latency (tools/microbench/gemm_lab stamps).
//   * issue stream and compute stream: one step of each per k-block; the issue stream runs DIST k-blocks ahead and
//     crosses into the next tile on its own (B pointers and the A pointers of the first time-offset slice are rebuilt
//     there).
//   * row maps are DMA'd into the SAME LDS arrays, each at the first issue step after its last reader: the next tile's
//     A row map at step (n_off - 1) * kb_per_off + 1 of a tile (the step before built the last slice's pointers), a
//     tile's own bypass row map at its step DIST (that step follows the barrier behind the previous tile's epilogue).
//     Either is complete before the next step's counted wait (it was requested before that step's k-block) and
//     visible after its barrier.  The host uses this kernel when kb_per_off >= 3 and nkb >= 6.
//   * epilogue scratch = the ring stage that held the tile's last k-block (free until the next tile's first step issues
//     into it): 16 KB, so a 32 x 32 accumulator tile goes through it in two halves of 16 rows.
//   * counted waits: every wait is vmcnt((DIST - 1) * LOADS).  Operations the count does not know about (row-map DMAs,
//     the epilogue's loads and stores) only ever make it wait longer: they are older than the k-block it leaves in
//     flight, or they are the stores of the epilogue before, all younger than the k-block it waits for.
template <int TI, int TJ, int DEPTH, bool IVB>
__device__ inline void EpilogueWaveHalves(const GemmArgs &p, f32x16 (&acc)[TI][TJ], int m_wave, int n_wave, int row_wave,
                                          float *scr /* [16][EPI_LD], this wave's */, const int *bm) {
  constexpr int NT = TI * TJ;
  const int lane = threadIdx.x & 63, c4 = (lane & 7) * 4, r8 = lane >> 3;
  const float *bias = p.bias ? p.bias : p.zeros_n, *bsc = p.bn_scale ? p.bn_scale : p.ones_n, *bof = p.bn_scale ? p.bn_offset : p.zeros_n;
  const float *pof = p.post_offset ? p.post_offset : p.zeros_n, *byp = p.byp ? p.byp : p.zeros_n;
  const int ld_byp = p.byp ? p.ld_byp : 0;
  const float byps = p.byp ? p.bypass_scale : 0.f, floor_ = p.relu ? 0.f : -INFINITY, posts = p.post_scale;
  auto request = [&](int t, EpiLoads &L) {
    const int i = t / TJ, j = t % TJ;
    const int n = n_wave + j * 32 + c4, nc = min(n, p.N - 4);
    L.bias = *reinterpret_cast<const float4 *>(bias + nc);
    L.bs = *reinterpret_cast<const float4 *>(bsc + nc);
    L.bo = *reinterpret_cast<const float4 *>(bof + nc);
    L.po = *reinterpret_cast<const float4 *>(pof + nc);
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
      const int row = row_wave + i * 32 + r8 + 8 * pass;
      L.z[pass] = *reinterpret_cast<const float4 *>(byp + static_cast<size_t>(bm[row]) * ld_byp + nc);
    }
  };
  EpiLoads L[DEPTH];
#pragma unroll
  for (int t = 0; t < DEPTH && t < NT; t++) request(t, L[t]);
  const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int i = t / TJ, j = t % TJ;
    const EpiLoads &E = L[t % DEPTH];
    const int n = n_wave + j * 32 + c4;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int r = 0; r < 8; r++) scr[((r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_LD + lr] = acc[i][j][8 * h + r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the scratch is private to the wave: no barrier
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float4 a[2];
#pragma unroll
      for (int q = 0; q < 2; q++) a[q] = *reinterpret_cast<const float4 *>(scr + (r8 + 8 * q) * EPI_LD + c4);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // the next half reuses the scratch
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int pass = 2 * h + q, m = m_wave + i * 32 + r8 + 8 * pass;
        float v[4] = {a[q].x + E.bias.x, a[q].y + E.bias.y, a[q].z + E.bias.z, a[q].w + E.bias.w};
        if (IVB) {
          const int mc = min(m, p.M - 1), nc = min(n, p.N - 4);
          const float4 iv = *reinterpret_cast<const float4 *>(p.ivbias + static_cast<size_t>(p.row2utt[mc]) * p.N + nc);
          v[0] += iv.x; v[1] += iv.y; v[2] += iv.z; v[3] += iv.w;
        }
        const float4 z = E.z[pass];
        v[0] = fmaxf(v[0], floor_); v[1] = fmaxf(v[1], floor_); v[2] = fmaxf(v[2], floor_); v[3] = fmaxf(v[3], floor_);
        v[0] = v[0] * E.bs.x + E.bo.x; v[1] = v[1] * E.bs.y + E.bo.y; v[2] = v[2] * E.bs.z + E.bo.z; v[3] = v[3] * E.bs.w + E.bo.w;
        v[0] += byps * z.x; v[1] += byps * z.y; v[2] += byps * z.z; v[3] += byps * z.w;
        v[0] += E.po.x; v[1] += E.po.y; v[2] += E.po.z; v[3] += E.po.w;
        if (m < p.M && n < p.N)
          *reinterpret_cast<float4 *>(p.C + static_cast<size_t>(m) * p.ldC + n) = make_float4(v[0] * posts, v[1] * posts, v[2] * posts, v[3] * posts);
      }
    }
    if (t + DEPTH < NT) request(t + DEPTH, L[t % DEPTH]);
  }
}

template <int BM, int BN, int WM, int WN, int NST, bool IVB, int OCC = 3, int DEPTH = 1>     // OCC workgroups per compute unit; DEPTH: tiles the epilogue's loads run ahead
__global__ __launch_bounds__(256, OCC) void TdnnGemmPersistKernel(GemmArgs p) {
  constexpr int BK = 16;
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM / 16, B_INST = BN / 16;
  constexpr int A_PW = (A_INST + 3) / 4, B_PW = (B_INST + 3) / 4;
  constexpr int LOADS = A_PW + B_PW;
  constexpr int DIST = NST - 1;
  static_assert(ST_BYTES >= 4 * 16 * EPI_LD * 4, "a ring stage doubles as the epilogue scratch of the four waves (half tiles)");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * ST_BYTES + (KAMD_MAX_OFFSETS + 1) * BM * 4];
  int *rm = reinterpret_cast<int *>(smem + NST * ST_BYTES);          // [n_off][BM]
  int *bm = rm + KAMD_MAX_OFFSETS * BM;                              // [BM]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int gx = p.gx, gy = p.gy;
  const int n_tiles = gx * ((gy + 7) & ~7);                          // XCD-aware order (see TdnnGemmKernel): some ids fall outside
  auto tile_by = [&](int id) { return (id & 7) + 8 * ((id >> 3) / gx); };
  auto tile_bx = [&](int id) { return (id >> 3) % gx; };
  auto next_tile = [&](int id) {                                     // the next id of this workgroup that is a tile
    id += gridDim.x;
    while (id < n_tiles && tile_by(id) >= gy) id += gridDim.x;
    return id;
  };
  int ct = static_cast<int>(blockIdx.x) - static_cast<int>(gridDim.x);
  ct = next_tile(ct);
  if (ct >= n_tiles) return;
  KAMD_STAMP(0); KAMD_STAMP_REAL(6);
  const int K = p.n_off * p.in_pad, nkb = K / BK, kb_per_off = p.in_pad / BK;
  // ---- the first tile's row maps the ordinary way
  {
    const int m0 = tile_by(ct) * BM;
    for (int i = t; i < p.n_off * BM; i += 256) {
      const int o = i / BM, r = i % BM;
      rm[o * BM + r] = p.rowmap[static_cast<size_t>(o) * p.M + min(m0 + r, p.M - 1)];
    }
    for (int r = t; r < BM; r += 256) bm[r] = p.byp ? p.bypmap[min(m0 + r, p.M - 1)] : 0;
  }
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)smem));
  const unsigned rm_lds = smem_lds + NST * ST_BYTES;
  auto dma16 = [&](const float *g, unsigned lds_addr) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(g) : "memory");
  };
  auto dma4 = [&](const int *g, unsigned lds_addr) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(g) : "memory");
  };
  const int lrow = lane >> 2, lslot = lane & 3;
  const float *pa[A_PW], *pb[B_PW];
  unsigned a_lds[A_PW], b_lds[B_PW];
#pragma unroll
  for (int q = 0; q < B_PW; q++) { int J = wave + 4 * q; if (J > B_INST - 1) J = B_INST - 1; b_lds[q] = A_BYTES + J * 1024; pb[q] = p.zeros; }
#pragma unroll
  for (int q = 0; q < A_PW; q++) { int I = wave + 4 * q; if (I > A_INST - 1) I = A_INST - 1; a_lds[q] = I * 1024; pa[q] = p.zeros; }
  // ---- issue stream
  int it = ct, ikb = 0;                       // tile and k-block the next issue step requests
  unsigned si = 0;                            // ring stage it goes to
  int in_flight = 0;                          // k-blocks requested and not yet multiplied
  const int map_step = (p.n_off - 1) * kb_per_off + 1;
  auto issue_step = [&]() {
    if (it >= n_tiles) return;
    if (ikb == 0) {
      const int n0 = tile_bx(it) * BN;
#pragma unroll
      for (int q = 0; q < B_PW; q++) {
        const int row = ((b_lds[q] - A_BYTES) >> 6) + lrow, c = lslot ^ ((row >> 2) & 3);
        pb[q] = p.W + static_cast<size_t>(n0 + row) * K + 4 * c;
      }
    }
    if (ikb % kb_per_off == 0) {              // uniform: a new time-offset slice starts
      const int off = ikb / kb_per_off;
#pragma unroll
      for (int q = 0; q < A_PW; q++) {
        const int row = (a_lds[q] >> 6) + lrow, c = lslot ^ ((row >> 2) & 3);
        const int src = rm[off * BM + row];
        pa[q] = p.A + static_cast<size_t>(src) * p.ldA + 4 * c;
      }
    }
    if (ikb == map_step) {                    // the NEXT tile's row maps, into the array nobody reads any more
      const int nt = next_tile(it);
      if (nt < n_tiles) {
        const int m0n = tile_by(nt) * BM;
        for (int i = wave; i < 2 * p.n_off; i += 4) {                  // pieces of 64 ints
          const int o = i >> 1, r = (i & 1) * 64 + lane;
          dma4(p.rowmap + static_cast<size_t>(o) * p.M + min(m0n + r, p.M - 1), rm_lds + (o * BM + (i & 1) * 64) * 4);
        }
      }
    }
    if (ikb == DIST && p.byp && wave < 2) {   // THIS tile's bypass rows: the epilogue of the tile before has just ended
      const int r = wave * 64 + lane;
      dma4(p.bypmap + min(tile_by(it) * BM + r, p.M - 1), rm_lds + (KAMD_MAX_OFFSETS * BM + wave * 64) * 4);
    }
    const unsigned st = smem_lds + si * ST_BYTES;
#pragma unroll
    for (int q = 0; q < A_PW; q++) { dma16(pa[q], st + a_lds[q]); pa[q] += BK; }
#pragma unroll
    for (int q = 0; q < B_PW; q++) { dma16(pb[q], st + b_lds[q]); pb[q] += BK; }
    si = si + 1 == NST ? 0 : si + 1;
    in_flight++;
    if (++ikb == nkb) { ikb = 0; it = next_tile(it); }
  };
#pragma unroll
  for (int d = 0; d < DIST; d++) issue_step();
  const int lr = lane & 31, lk = lane >> 5;
  unsigned sc = 0;                            // ring stage the next compute step reads
  while (ct < n_tiles) {
    const int m0 = tile_by(ct) * BM, n0 = tile_bx(ct) * BN;
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int kb = 0; kb < nkb; kb++) {
      // the k-block about to be multiplied has landed once all but the k-blocks requested after it are done
#ifdef KAMD_GEMM_LAB
      if (p.lab_prio) __builtin_amdgcn_s_setprio(3);
#endif
      if (in_flight >= DIST) {
        if (DIST >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
        else if (DIST == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (DIST >= 3 && in_flight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      in_flight--;
      issue_step();                           // into the stage everybody finished reading before this barrier
#ifdef KAMD_GEMM_LAB
      if (p.lab_prio) __builtin_amdgcn_s_setprio(0);
#endif
      const unsigned char *st = smem + sc * ST_BYTES;
      sc = sc + 1 == NST ? 0 : sc + 1;
      float4 a[TI][2], b[TJ][2];
#pragma unroll
      for (int i = 0; i < TI; i++) {
        const int m = wm * (BM / WM) + i * 32 + lr;
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
          a[i][tt] = *reinterpret_cast<const float4 *>(st + m * 64 + (((2 * tt + lk) ^ ((m >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        const int n = wn * (BN / WN) + j * 32 + lr;
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
          b[j][tt] = *reinterpret_cast<const float4 *>(st + A_BYTES + n * 64 + (((2 * tt + lk) ^ ((n >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int i = 0; i < TI; i++) {
            const float av = q == 0 ? a[i][tt].x : q == 1 ? a[i][tt].y : q == 2 ? a[i][tt].z : a[i][tt].w;
#pragma unroll
            for (int j = 0; j < TJ; j++) {
              const float bv = q == 0 ? b[j][tt].x : q == 1 ? b[j][tt].y : q == 2 ? b[j][tt].z : b[j][tt].w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_barrier();            // every wave has read the tile's last stage: it is the epilogue's scratch now
    asm volatile("" ::: "memory");
    const unsigned s_last = sc == 0 ? NST - 1 : sc - 1;
    float *scr = reinterpret_cast<float *>(smem + s_last * ST_BYTES) + wave * 16 * EPI_LD;
    EpilogueWaveHalves<TI, TJ, DEPTH, IVB>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), wm * (BM / WM), scr, bm);
    asm volatile("" ::: "memory");
    ct = next_tile(ct);
  }
  KAMD_STAMP(5); KAMD_STAMP_REAL(7);
}

// LogSoftmaxComponent::Propagate = ApplyLogSoftMaxPerRow (nnet-simple-component.cc:3599;
// matrix/kaldi-vector.cc:876-884) fused with DecodableNnetSimple's "-log_prior, *scale"
// (nnet-am-decodable-simple.cc:268-271): one wavefront per row, the row (<= 24 KB) was just
// written by the GEMM and is read back from L2.  HBM-trivial: 2 x 4 B per element.
__global__ __launch_bounds__(256) void LogSoftmaxRowsKernel(float *C, int ldC, int M, int N,
                                                            const float *post_offset, float post_scale) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float *x = C + static_cast<size_t>(row) * ldC;
  float mx = -INFINITY;
  for (int n = lane; n < N; n += 64) mx = fmaxf(mx, x[n]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int n = lane; n < N; n += 64) sum += expf(x[n] - mx);
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float lse = logf(sum);
  for (int n = lane; n < N; n += 64) {
    float v = (x[n] - mx) - lse;
    if (post_offset) v += post_offset[n];
    x[n] = v * post_scale;
  }
}

// Per-utterance time grids -> row maps.  Row r of layer l (utterance u, k-th row) is
// time t = lo_l + k*step_l; its operand for offset o lives at producer row
// (t + o - lo_p)/step_p of the same utterance, or, for the network input, at feature
// row clamp(t + o, 0, T_u - 1) (edge clamping of nnet-am-decodable-simple.cc:147-160).
struct MapArgs {
  int n_utts, M;
  const int64_t *row_off;       // [n_utts+1] rows of this layer
  const int64_t *prod_row_off;  // [n_utts+1] rows of the producer (or feature rows)
  const int *T;                 // [n_utts] input frames
  int lo, step, prod_lo, prod_step, prod_is_input;
  int n_off; int offs[KAMD_MAX_OFFSETS];
  int *rowmap;                  // [n_off][M]
  int *row2utt;                 // [M] or NULL
  // per-row i-vector slots (the looped decodable's Round(ivector, period)): row2utt[r] then indexes a table
  // of i-vectors, slot_base[u] + clamp(floor((abs_t0[u] + t) / period) - slot_first[u], 0, slot_count[u] - 1)
  int slot_period;              // 0: one i-vector per item
  const int *slot_base, *slot_first, *slot_count, *abs_t0;
  // items that are CHUNKS of an utterance (DecodableNnetSimple with online i-vectors): the item's time 0 is input frame
  // in_t0[u] of the utterance whose rows start at prod_row_off[u] and whose T[u] frames bound the clamp.  NULL: 0.
  const int *in_t0;
};
__device__ inline void RowMapRow(const MapArgs &a, int r) {
  if (r >= a.M) return;
  int lo = 0, hi = a.n_utts;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (a.row_off[mid] <= r) lo = mid; else hi = mid;
  }
  const int u = lo;
  const int t = a.lo + static_cast<int>(r - a.row_off[u]) * a.step;
  if (a.row2utt) {
    if (a.slot_period > 0) {
      const int ta = a.abs_t0[u] + t;
      int sl = (ta >= 0 ? ta / a.slot_period : -((-ta + a.slot_period - 1) / a.slot_period)) - a.slot_first[u];
      sl = sl < 0 ? 0 : (sl >= a.slot_count[u] ? a.slot_count[u] - 1 : sl);
      a.row2utt[r] = a.slot_base[u] + sl;
    } else a.row2utt[r] = u;
  }
  for (int i = 0; i < a.n_off; i++) {
    int tin = t + a.offs[i];
    int64_t src;
    if (a.prod_is_input) {
      if (a.in_t0) tin += a.in_t0[u];
      int c = tin < 0 ? 0 : (tin >= a.T[u] ? a.T[u] - 1 : tin);
      src = a.prod_row_off[u] + c;
    } else {
      src = a.prod_row_off[u] + (tin - a.prod_lo) / a.prod_step;
    }
    a.rowmap[static_cast<size_t>(i) * a.M + r] = static_cast<int>(src);
  }
}
// Every row map of a pass in ONE launch (the maps depend on the pass's descriptors only, not on activations): map k owns
// the blocks [block_start[k], block_start[k + 1]).  89 GEMMs of a pass used to come with 144 map launches, which is what a
// streaming tick of a few hundred rows spends its time on.
__global__ __launch_bounds__(256) void RowMapAllKernel(const MapArgs *maps, const int *block_start, int n_maps) {
  const int b = blockIdx.x;
  int lo = 0, hi = n_maps;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (block_start[mid] <= b) lo = mid; else hi = mid;
  }
  const MapArgs a = maps[lo];
  RowMapRow(a, (b - block_start[lo]) * 256 + static_cast<int>(threadIdx.x));
}

// one slice of a concat layer: dst[r][0 .. dim) = src[map[r]][0 .. dim)
__global__ void CopySliceKernel(const float *src, int ld_src, const int *map, float *dst, int ld_dst, int dim, int M) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const float *s = src + static_cast<size_t>(map[r]) * ld_src;
  float *d = dst + static_cast<size_t>(r) * ld_dst;
  for (int k = threadIdx.x & 63; k < dim; k += 64) d[k] = s[k];
}

// ivbias[u][n] = sum_k W_iv[n][k] * ivec[u][k]   (ReplaceIndex(ivector, t, 0) operand)
// (rows != NULL: item u reads row rows[u] of the i-vector table)
__global__ void IvecBiasKernel(const float *Wiv, const float *ivec, int N, int D, float *out, const int *rows) {
  int n = blockIdx.y * blockDim.x + threadIdx.x, u = blockIdx.x;     // (items on x: there can be more than 65535)
  if (n >= N) return;
  const size_t src = rows ? static_cast<size_t>(rows[u]) : static_cast<size_t>(u);
  float s = 0.f;
  for (int k = 0; k < D; k++) s += Wiv[static_cast<size_t>(n) * D + k] * ivec[src * D + k];
  out[static_cast<size_t>(u) * N + n] = s;
}

// ------------------------------------------------------------------ host
struct LayerDev {
  int in_dim, out_dim, in_pad, out_pad, n_off, offs[KAMD_MAX_OFFSETS];
  int input_layer, bypass_layer, ivector_dim, relu, log_softmax;
  float bypass_scale, post_scale;
  float *W = NULL, *Wiv = NULL, *bias = NULL, *bn_scale = NULL, *bn_offset = NULL, *post_offset = NULL;
  int N_pad;
  // time grid (per layer constants): rows for utterance u are t = lo + k*step,
  // k < (hi_const + 3*(n_out_u-1) - lo)/step + 1
  int lo, hi_const, step;
  // A CONCAT pseudo-layer (no GEMM): the materialised Append over different producers in front of a multi_input layer
  // (what nnet3 does with kCopyRows, nnet-compute.cc:309-383): slice o = rows of layer sl_layer[o] (-1: the input) at
  // time offset offs[o], sl_dim[o] columns, written at column sl_col[o] of this layer's rows
  bool concat = false;
  int sl_layer[KAMD_MAX_OFFSETS], sl_dim[KAMD_MAX_OFFSETS], sl_col[KAMD_MAX_OFFSETS];
};

struct Nnet {
  std::vector<LayerDev> L;
  int input_dim, subsampling, left, right;
  int in_lo, in_hi_const;  // time range of the network input that is touched
  double last_flops = 0;
  // batch workspace (grown on demand)
  std::vector<float *> act; std::vector<size_t> act_cap;
  std::vector<int *> maps; std::vector<size_t> maps_cap;
  kamd::MetaRing meta;       // per-pass descriptors (row offsets of every layer, item lengths ...): meta_ring.h
  float *d_ivb = NULL; size_t ivb_cap = 0;
  float *d_zero = NULL;      // n_neutral zero floats (GemmArgs::zeros, zeros_n) followed by n_neutral ones (ones_n)
  float *d_scratch = NULL; size_t scratch_cap = 0;     // kamd::NnetScratch: the streaming entry's temporary rows
  int n_neutral = 0;
};

static int Mod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

// Required time set of every layer as an arithmetic progression (lo, hi, step) with
// hi = hi_const + sub*(n_out-1); step is 'sub' iff all requirements share one residue.
static int PlanGrids(Nnet *nn) {
  int n = static_cast<int>(nn->L.size()), sub = nn->subsampling;
  struct Req { bool any; int lo, hi; int residue; bool dense; };
  std::vector<Req> req(n + 1);  // index n = network input
  for (int i = 0; i <= n; i++) { req[i].any = false; req[i].dense = false; req[i].residue = -1; req[i].lo = 0; req[i].hi = 0; }
  auto add = [&](int idx, int lo, int hi, int residue, bool dense) {
    Req &r = req[idx];
    if (!r.any) { r.any = true; r.lo = lo; r.hi = hi; r.residue = residue; r.dense = dense; return; }
    r.lo = std::min(r.lo, lo); r.hi = std::max(r.hi, hi);
    if (dense || r.residue != residue) r.dense = true;
  };
  add(n - 1, 0, 0, 0, sub == 1);  // output node: t = 0, sub, 2 sub, ... (hi const part 0)
  for (int i = n - 1; i >= 0; i--) {
    LayerDev &l = nn->L[i];
    Req &r = req[i];
    if (!r.any) return SetError(KAMD_ERR_ARG, "layer %d has no consumer", i);
    l.step = r.dense ? 1 : sub;
    l.lo = r.lo; l.hi_const = r.hi;
    if (!r.dense) {  // snap lo/hi onto the residue class
      // lo, hi already lie on the residue class by construction
    }
    for (int o = 0; o < l.n_off; o++) {
      const int prod = l.concat ? l.sl_layer[o] : l.input_layer;
      const int pidx = prod < 0 ? n : prod;
      int off = l.offs[o];
      add(pidx, r.lo + off, r.hi + off, r.dense ? -1 : Mod(r.residue + off, sub), r.dense);
    }
    if (l.bypass_layer != -2) {
      int bidx = l.bypass_layer < 0 ? n : l.bypass_layer;
      add(bidx, r.lo, r.hi, r.dense ? -1 : r.residue, r.dense);
    }
  }
  nn->in_lo = req[n].lo; nn->in_hi_const = req[n].hi;
  nn->left = -req[n].lo; nn->right = req[n].hi;
  return KAMD_OK;
}

static inline int64_t LayerRows(const LayerDev &l, int n_out, int sub) {
  int hi = l.hi_const + sub * (n_out - 1);
  return (hi - l.lo) / l.step + 1;
}

// The zero fill is issued on the stream that will use the buffer: a hipMemset on the NULL stream is not ordered with
// kernels on a non-blocking stream, and a fill that is still running when the first GEMM writes the buffer wipes its
// output (seen once another kernel kept compute units busy and the fill kernel started late).
template <typename T>
static int Grow(T **p, size_t *cap, size_t need, hipStream_t st) {
  if (need <= *cap) return KAMD_OK;
  if (*p) { KAMD_HIP(hipStreamSynchronize(st)); (void)hipFree(*p); }
  *p = NULL;
  size_t n = need + need / 4;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
  KAMD_HIP(hipMemsetAsync(*p, 0, n * sizeof(T), st));
  *cap = n;
  return KAMD_OK;
}

// A grow-only device buffer owned by the model, for callers that need a few temporary rows per call (kamd_nnet_forward_range
// after every chunk of a stream: hipMalloc / hipFree synchronise the device).  Not thread-safe, like the model's workspaces.
float *NnetScratch(kamd_nnet *h, size_t floats) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  if (floats > nn->scratch_cap) {
    if (nn->d_scratch) (void)hipFree(nn->d_scratch);
    nn->d_scratch = NULL; nn->scratch_cap = 0;
    const size_t cap = floats + floats / 2;
    if (hipMalloc(reinterpret_cast<void **>(&nn->d_scratch), cap * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    nn->scratch_cap = cap;
  }
  return nn->d_scratch;
}

}  // namespace kamd

using kamd::Nnet;
using kamd::LayerDev;

extern "C" {

kamd_nnet *kamd_nnet_create(const kamd_layer_desc *layers, int n_layers, int input_dim,
                            int subsampling) {
  if (n_layers < 1 || subsampling < 1) { kamd::SetError(KAMD_ERR_ARG, "bad nnet description"); return NULL; }
  if (!kamd::RequireDevice()) return NULL;
  Nnet *nn = new Nnet();
  nn->input_dim = input_dim; nn->subsampling = subsampling;
  // The device's layer list: the caller's layers, with a CONCAT pseudo-layer in front of every multi_input one (the
  // caller's indices are mapped through ext2int).
  nn->L.reserve(2 * static_cast<size_t>(n_layers));
  std::vector<int> ext2int(n_layers, -1);
  auto remap = [&](int x) { return x < 0 ? x : ext2int[x]; };
  for (int i = 0; i < n_layers; i++) {
    const kamd_layer_desc &s = layers[i];
    if (s.n_offsets < 1 || s.n_offsets > KAMD_MAX_OFFSETS || s.input_layer >= i || s.bypass_layer >= i ||
        (s.ivector_dim > 0 && (s.input_layer != -1 || s.multi_input))) {
      kamd::SetError(KAMD_ERR_ARG, "layer %d: bad topology", i);
      delete nn; return NULL;
    }
    int n_off_l = s.n_offsets, input_l = remap(s.input_layer);
    if (s.multi_input) {
      LayerDev cat;
      cat.concat = true; cat.n_off = s.n_offsets; cat.in_dim = 0; cat.in_pad = 0; cat.input_layer = -3; cat.bypass_layer = -2; cat.ivector_dim = 0;
      cat.relu = 0; cat.log_softmax = 0; cat.bypass_scale = 0.0f; cat.post_scale = 1.0f; cat.N_pad = 0;
      int width = 0;
      for (int o = 0; o < s.n_offsets; o++) {
        const int pl = s.slice_layer[o];
        if (pl >= i || pl < -1 || s.slice_dim[o] != (pl < 0 ? input_dim : layers[pl].out_dim)) {
          kamd::SetError(KAMD_ERR_ARG, "layer %d: slice %d does not match its producer", i, o);
          delete nn; return NULL;
        }
        cat.offs[o] = s.offsets[o]; cat.sl_layer[o] = remap(pl); cat.sl_dim[o] = s.slice_dim[o]; cat.sl_col[o] = width;
        width += s.slice_dim[o];
      }
      if (width != s.in_dim) { kamd::SetError(KAMD_ERR_ARG, "layer %d: in_dim %d != the sum of its slices %d", i, s.in_dim, width); delete nn; return NULL; }
      cat.out_dim = width; cat.out_pad = kamd::RoundUp(width, 16);
      nn->L.push_back(cat);
      n_off_l = 1; input_l = static_cast<int>(nn->L.size()) - 1;        // the affine part reads the concatenation at offset 0
    } else {
      int prod_dim = s.input_layer < 0 ? input_dim : layers[s.input_layer].out_dim;
      if (prod_dim != s.in_dim) { kamd::SetError(KAMD_ERR_ARG, "layer %d: in_dim %d != producer dim %d", i, s.in_dim, prod_dim); delete nn; return NULL; }
    }
    if (s.bypass_layer != -2) {
      int bd = s.bypass_layer < 0 ? input_dim : layers[s.bypass_layer].out_dim;
      if (bd != s.out_dim) { kamd::SetError(KAMD_ERR_ARG, "layer %d: bypass dim mismatch", i); delete nn; return NULL; }
    }
    nn->L.push_back(LayerDev());
    LayerDev &l = nn->L.back();
    ext2int[i] = static_cast<int>(nn->L.size()) - 1;
    l.in_dim = s.in_dim; l.out_dim = s.out_dim; l.in_pad = kamd::RoundUp(s.in_dim, 16);
    l.out_pad = kamd::RoundUp(s.out_dim, 16); l.n_off = n_off_l;
    for (int o = 0; o < n_off_l; o++) l.offs[o] = s.multi_input ? 0 : s.offsets[o];
    l.input_layer = input_l; l.bypass_layer = s.bypass_layer == -2 ? -2 : remap(s.bypass_layer); l.ivector_dim = s.ivector_dim;
    l.relu = s.relu; l.log_softmax = s.log_softmax; l.bypass_scale = s.bypass_scale; l.post_scale = s.post_scale;
    l.N_pad = kamd::RoundUp(s.out_dim, 128);
    const int Ksrc = n_off_l * s.in_dim + s.ivector_dim, Kp = n_off_l * l.in_pad;
    std::vector<float> Wp(static_cast<size_t>(l.N_pad) * Kp, 0.0f);
    for (int n = 0; n < s.out_dim; n++)
      for (int o = 0; o < n_off_l; o++)
        memcpy(&Wp[static_cast<size_t>(n) * Kp + o * l.in_pad], s.W + static_cast<size_t>(n) * Ksrc + o * s.in_dim,
               sizeof(float) * s.in_dim);
    auto up = [&](const float *src, size_t cnt) -> float * {
      float *d = kamd::DevAlloc<float>(cnt);
      if (d && hipMemcpy(d, src, cnt * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
      return d;
    };
    l.W = up(Wp.data(), Wp.size());
    bool ok = l.W != NULL;
    if (s.ivector_dim > 0) {
      std::vector<float> wiv(static_cast<size_t>(s.out_dim) * s.ivector_dim);
      for (int n = 0; n < s.out_dim; n++)
        memcpy(&wiv[static_cast<size_t>(n) * s.ivector_dim], s.W + static_cast<size_t>(n) * Ksrc + n_off_l * s.in_dim,
               sizeof(float) * s.ivector_dim);
      l.Wiv = up(wiv.data(), wiv.size()); ok = ok && l.Wiv;
    }
    if (s.bias) { l.bias = up(s.bias, s.out_dim); ok = ok && l.bias; }
    if (s.bn_scale) { l.bn_scale = up(s.bn_scale, s.out_dim); l.bn_offset = up(s.bn_offset, s.out_dim); ok = ok && l.bn_scale && l.bn_offset; }
    if (s.post_offset) { l.post_offset = up(s.post_offset, s.out_dim); ok = ok && l.post_offset; }
    if (!ok) { kamd::SetError(KAMD_ERR_HIP, "weight upload failed (layer %d)", i); delete nn; return NULL; }
  }
  if (kamd::PlanGrids(nn) != KAMD_OK) { delete nn; return NULL; }
  n_layers = static_cast<int>(nn->L.size());          // from here on: the device's list
  int max_n = 64;
  for (int i = 0; i < n_layers; i++) max_n = std::max(max_n, nn->L[i].N_pad);
  nn->n_neutral = max_n;
  {
    std::vector<float> neutral(2 * static_cast<size_t>(max_n), 0.0f);
    std::fill(neutral.begin() + max_n, neutral.end(), 1.0f);
    nn->d_zero = kamd::DevAlloc<float>(neutral.size());
    if (!nn->d_zero || hipMemcpy(nn->d_zero, neutral.data(), neutral.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) { kamd::SetError(KAMD_ERR_HIP, "nnet: allocation failed"); delete nn; return NULL; }
  }
  nn->act.assign(n_layers, NULL); nn->act_cap.assign(n_layers, 0);
  nn->maps.assign(n_layers, NULL); nn->maps_cap.assign(n_layers, 0);
  return reinterpret_cast<kamd_nnet *>(nn);
}

void kamd_nnet_destroy(kamd_nnet *h) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  if (!nn) return;
  for (size_t i = 0; i < nn->L.size(); i++) {
    LayerDev &l = nn->L[i];
    float *ps[] = {l.W, l.Wiv, l.bias, l.bn_scale, l.bn_offset, l.post_offset};
    for (float *p : ps) if (p) (void)hipFree(p);
    if (nn->act[i]) (void)hipFree(nn->act[i]);
    if (nn->maps[i]) (void)hipFree(nn->maps[i]);
  }
  if (nn->d_ivb) (void)hipFree(nn->d_ivb);
  if (nn->d_zero) (void)hipFree(nn->d_zero);
  if (nn->d_scratch) (void)hipFree(nn->d_scratch);
  delete nn;
}

int kamd_nnet_output_dim(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->L.back().out_dim; }
int kamd_nnet_input_dim(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->input_dim; }
int kamd_nnet_ivector_dim(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->L[0].ivector_dim; }
int kamd_nnet_left_context(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->left; }
int kamd_nnet_right_context(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->right; }
int kamd_nnet_frame_subsampling_factor(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->subsampling; }
int kamd_nnet_num_output_frames(const kamd_nnet *h, int T) {
  int s = reinterpret_cast<const Nnet *>(h)->subsampling;
  return (T + s - 1) / s;
}
double kamd_nnet_last_flops(const kamd_nnet *h) { return reinterpret_cast<const Nnet *>(h)->last_flops; }

// h_in_start[u] / h_in_len[u]: first feature row and number of input frames of item u (the
// items need not be adjacent: streaming slices of different streams live in one pooled buffer)
struct SlotSpec {                 // host arrays, one entry per item; table rows = sum of slot_count
  int period, table_rows;
  const int32_t *slot_base, *slot_first, *slot_count, *abs_t0;
};
// Items that are chunks of utterances: item u produces n_out[u] output frames starting at input frame t0[u] (a multiple
// of the subsampling factor) of the utterance h_in_start[u] / h_in_len[u] describe (the WHOLE utterance: its edges are
// where the input is clamped); every layer is evaluated at exactly the times the chunk's outputs need.
struct ChunkSpec { const int32_t *t0, *n_out, *iv_row; };    // iv_row[u]: the row of d_ivectors item u reads
static int ForwardItems(kamd_nnet *h, const float *d_feats, const int64_t *h_in_start, const int32_t *h_in_len,
                        int ld_in, const float *d_ivectors, int n_utts, float *d_out,
                        const int64_t *h_out_row_off, int ld_out, void *stream, const SlotSpec *slots = NULL,
                        const ChunkSpec *chunks = NULL) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nl = static_cast<int>(nn->L.size()), sub = nn->subsampling;
  if (n_utts <= 0) return KAMD_OK;
  if (ld_in < nn->L[0].in_pad && nn->L[0].input_layer == -1)
    return kamd::SetError(KAMD_ERR_ARG, "ld_in %d < padded input dim %d (pad columns must be zero)", ld_in, nn->L[0].in_pad);
  if (ld_out < nn->L.back().out_dim) return kamd::SetError(KAMD_ERR_ARG, "ld_out too small");
  // ---- meta: T[u] (as int64 for simplicity), per-layer row offsets
  // layout: [ (nl+2) arrays of (n_utts+1) int64 ] then T as int32
  const size_t stride = n_utts + 1;
  const size_t t_words = (n_utts + 1) / 2 + 1, slot_words = slots ? static_cast<size_t>(2 * n_utts + 2) : 0;
  const size_t chunk_words = chunks ? 2 * t_words : 0;
  std::vector<int64_t> meta((nl + 2) * stride + t_words + slot_words + chunk_words, 0);
  int *Th = reinterpret_cast<int *>(&meta[(nl + 2) * stride]);
  int *Sh = reinterpret_cast<int *>(&meta[(nl + 2) * stride + t_words]);      // slot_base | slot_first | slot_count | abs_t0
  int *Ch = reinterpret_cast<int *>(&meta[(nl + 2) * stride + t_words + slot_words]);   // t0 of chunk items
  if (slots)
    for (int u = 0; u < n_utts; u++) {
      Sh[u] = slots->slot_base[u]; Sh[n_utts + u] = slots->slot_first[u];
      Sh[2 * n_utts + u] = slots->slot_count[u]; Sh[3 * n_utts + u] = slots->abs_t0[u];
      if (slots->slot_count[u] <= 0 || slots->slot_base[u] < 0 || slots->slot_base[u] + slots->slot_count[u] > slots->table_rows)
        return kamd::SetError(KAMD_ERR_ARG, "item %d: bad i-vector slot range", u);
    }
  std::vector<int> n_out(n_utts);
  for (int u = 0; u < n_utts; u++) {
    int T = h_in_len[u];
    if (T <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames", u);
    Th[u] = T; n_out[u] = (T + sub - 1) / sub;
    if (chunks) {
      if (chunks->n_out[u] <= 0 || chunks->t0[u] < 0 || chunks->t0[u] % sub != 0 || chunks->t0[u] / sub + chunks->n_out[u] > n_out[u])
        return kamd::SetError(KAMD_ERR_ARG, "item %d: bad chunk", u);
      n_out[u] = chunks->n_out[u]; Ch[u] = chunks->t0[u]; Ch[2 * t_words + u] = chunks->iv_row[u];
    }
    meta[nl * stride + u] = h_in_start[u];           // feature rows
  }
  meta[nl * stride + n_utts] = h_in_start[n_utts - 1] + h_in_len[n_utts - 1];
  std::vector<int64_t> M(nl, 0);
  for (int l = 0; l < nl; l++) {
    int64_t acc = 0;
    for (int u = 0; u < n_utts; u++) {
      // the last layer writes straight into the caller's buffer at its row offsets
      meta[l * stride + u] = (l == nl - 1) ? h_out_row_off[u] : acc;
      acc += kamd::LayerRows(nn->L[l], n_out[u], sub);
    }
    meta[l * stride + n_utts] = (l == nl - 1) ? h_out_row_off[n_utts - 1] + n_out[n_utts - 1] : acc;
    M[l] = acc;
    if (acc > 2000000000LL) return kamd::SetError(KAMD_ERR_ARG, "batch too large");
  }
  // the output layer's rows must be contiguous in the caller's buffer for the map
  // kernel's binary search: require h_out_row_off to be the running sum of n_out.
  for (int u = 0; u + 1 < n_utts; u++)
    if (h_out_row_off[u + 1] != h_out_row_off[u] + n_out[u])
      return kamd::SetError(KAMD_ERR_ARG, "h_out_row_off must be the running sum of output frames");
  // the last layer's row offsets are absolute rows of d_out; the map kernel works in layer-local rows: a relative copy
  for (size_t i = 0; i < stride; i++) meta[(nl + 1) * stride + i] = meta[(nl - 1) * stride + i] - h_out_row_off[0];
  // ---- workspaces of every layer
  struct LayerBufs { float *C; int ldC; int64_t row_base; int *rowmap, *bypmap, *row2utt; };
  std::vector<LayerBufs> bufs(nl);
  int n_maps = 0;
  for (int l = 0; l < nl; l++) {
    LayerDev &L = nn->L[l];
    LayerBufs &B = bufs[l];
    B.row_base = 0;
    if (l == nl - 1) { B.C = d_out; B.ldC = ld_out; B.row_base = h_out_row_off[0]; }
    else {
      if (kamd::Grow(&nn->act[l], &nn->act_cap[l], static_cast<size_t>(M[l]) * L.out_pad, st) != KAMD_OK) return KAMD_ERR_HIP;
      B.C = nn->act[l]; B.ldC = L.out_pad;
    }
    if (kamd::Grow(&nn->maps[l], &nn->maps_cap[l], static_cast<size_t>(L.n_off + 2) * M[l], st) != KAMD_OK) return KAMD_ERR_HIP;
    B.rowmap = nn->maps[l]; B.bypmap = B.rowmap + static_cast<size_t>(L.n_off) * M[l]; B.row2utt = B.bypmap + M[l];
    n_maps += L.concat ? L.n_off : 1 + (L.bypass_layer != -2 ? 1 : 0);
  }
  // ---- the descriptors go to the device behind this stream's work, with no host wait (meta_ring.h): the int64 words
  // above, then one MapArgs per row map (they point into the same buffer), then the maps' first blocks
  const size_t maps_at = meta.size() * 8, starts_at = maps_at + static_cast<size_t>(n_maps) * sizeof(kamd::MapArgs);
  void *h_meta_v = NULL, *d_meta_v = NULL;
  if (nn->meta.Reserve(starts_at + static_cast<size_t>(n_maps + 1) * sizeof(int), &h_meta_v, &d_meta_v) != KAMD_OK) return KAMD_ERR_HIP;
  struct Releaser { kamd::MetaRing &m; hipStream_t s; ~Releaser() { (void)m.Release(s); } } releaser{nn->meta, st};
  memcpy(h_meta_v, meta.data(), maps_at);
  kamd::MapArgs *h_maps = reinterpret_cast<kamd::MapArgs *>(static_cast<char *>(h_meta_v) + maps_at);
  int *h_starts = reinterpret_cast<int *>(static_cast<char *>(h_meta_v) + starts_at);
  const int64_t *const d_meta = static_cast<const int64_t *>(d_meta_v);
  const int *d_T = reinterpret_cast<const int *>(d_meta + (nl + 2) * stride);
  int km = 0;
  int64_t map_blocks = 0;
  auto add_map = [&](const kamd::MapArgs &a) {
    h_maps[km] = a; h_starts[km] = static_cast<int>(map_blocks); km++;
    map_blocks += kamd::CeilDiv(a.M, 256);
  };
  for (int l = 0; l < nl; l++) {
    LayerDev &L = nn->L[l];
    const LayerBufs &B = bufs[l];
    const int64_t Ml = M[l];
    kamd::MapArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.n_utts = n_utts; ma.M = static_cast<int>(Ml);
    ma.T = d_T; ma.lo = L.lo; ma.step = L.step;
    ma.row2utt = L.ivector_dim > 0 ? B.row2utt : NULL;
    ma.slot_period = 0; ma.slot_base = ma.slot_first = ma.slot_count = ma.abs_t0 = NULL;
    ma.in_t0 = chunks ? reinterpret_cast<const int *>(d_meta + (nl + 2) * stride + t_words + slot_words) : NULL;
    if (slots && L.ivector_dim > 0) {
      const int *d_S = reinterpret_cast<const int *>(d_meta + (nl + 2) * stride + t_words);
      ma.slot_period = slots->period;
      ma.slot_base = d_S; ma.slot_first = d_S + n_utts; ma.slot_count = d_S + 2 * n_utts; ma.abs_t0 = d_S + 3 * n_utts;
    }
    auto fill_prod = [&](int prod, kamd::MapArgs *a) {
      if (prod < 0) { a->prod_row_off = d_meta + nl * stride; a->prod_is_input = 1; a->prod_lo = 0; a->prod_step = 1; }
      else { a->prod_row_off = d_meta + prod * stride; a->prod_is_input = 0; a->prod_lo = nn->L[prod].lo; a->prod_step = nn->L[prod].step; }
    };
    // the last layer's row offsets are absolute rows of d_out; the map kernel works in layer-local rows: the relative copy
    ma.row_off = (l == nl - 1 && B.row_base != 0) ? d_meta + (nl + 1) * stride : d_meta + l * stride;
    if (L.concat) {
      // the materialised Append: per slice a row map into its own producer (the gather-copy follows in the layer loop)
      for (int o = 0; o < L.n_off; o++) {
        kamd::MapArgs ms = ma;
        fill_prod(L.sl_layer[o], &ms);
        ms.n_off = 1; ms.offs[0] = L.offs[o]; ms.rowmap = B.rowmap + static_cast<size_t>(o) * Ml; ms.row2utt = NULL;
        add_map(ms);
      }
      continue;
    }
    fill_prod(L.input_layer, &ma);
    ma.n_off = L.n_off;
    for (int o = 0; o < L.n_off; o++) ma.offs[o] = L.offs[o];
    ma.rowmap = B.rowmap;
    add_map(ma);
    if (L.bypass_layer != -2) {
      kamd::MapArgs mb = ma;
      fill_prod(L.bypass_layer, &mb);
      mb.n_off = 1; mb.offs[0] = 0; mb.rowmap = B.bypmap; mb.row2utt = NULL;
      add_map(mb);
    }
  }
  h_starts[km] = static_cast<int>(map_blocks);
  if (km != n_maps || map_blocks > 2000000000LL) return kamd::SetError(KAMD_ERR_STATE, "row map plan");
  if (nn->meta.Commit(st) != KAMD_OK) return KAMD_ERR_HIP;
  if (map_blocks > 0)
    hipLaunchKernelGGL(kamd::RowMapAllKernel, dim3(static_cast<unsigned>(map_blocks)), dim3(256), 0, st,
                       reinterpret_cast<const kamd::MapArgs *>(static_cast<const char *>(d_meta_v) + maps_at),
                       reinterpret_cast<const int *>(static_cast<const char *>(d_meta_v) + starts_at), n_maps);
  KAMD_HIP(hipGetLastError());
  double flops = 0;
  for (int l = 0; l < nl; l++) {
    LayerDev &L = nn->L[l];
    const int64_t Ml = M[l];
    float *const C = bufs[l].C; const int ldC = bufs[l].ldC;
    const int64_t row_base = bufs[l].row_base;
    int *const rowmap = bufs[l].rowmap, *const bypmap = bufs[l].bypmap, *const row2utt = bufs[l].row2utt;
    if (L.concat) {
      for (int o = 0; o < L.n_off; o++) {
        const float *src = L.sl_layer[o] < 0 ? d_feats : nn->act[L.sl_layer[o]];
        const int ld_src = L.sl_layer[o] < 0 ? ld_in : nn->L[L.sl_layer[o]].out_pad;
        hipLaunchKernelGGL(kamd::CopySliceKernel, dim3(kamd::CeilDiv(Ml, 4)), dim3(256), 0, st, src, ld_src, rowmap + static_cast<size_t>(o) * Ml,
                           C + L.sl_col[o], ldC, L.sl_dim[o], static_cast<int>(Ml));
      }
      KAMD_HIP(hipGetLastError());
      continue;
    }
    kamd::GemmArgs g;
    memset(&g, 0, sizeof(g));
    if (L.input_layer < 0) { g.A = d_feats; g.ldA = ld_in; }
    else { g.A = nn->act[L.input_layer]; g.ldA = nn->L[L.input_layer].out_pad; }
    g.rowmap = rowmap; g.M = static_cast<int>(Ml); g.N = L.out_dim; g.n_off = L.n_off; g.in_pad = L.in_pad;
    g.W = L.W; g.bias = L.bias; g.relu = L.relu; g.bn_scale = L.bn_scale; g.bn_offset = L.bn_offset;
    if (L.ivector_dim > 0) {
      if (!d_ivectors) return kamd::SetError(KAMD_ERR_ARG, "model needs ivectors");
      const int iv_rows = slots ? slots->table_rows : n_utts;
      if (kamd::Grow(&nn->d_ivb, &nn->ivb_cap, static_cast<size_t>(iv_rows) * L.out_dim, st) != KAMD_OK) return KAMD_ERR_HIP;
      const int *d_iv_rows = chunks ? reinterpret_cast<const int *>(d_meta + (nl + 2) * stride + t_words + slot_words) + 2 * t_words : NULL;
      hipLaunchKernelGGL(kamd::IvecBiasKernel, dim3(iv_rows, kamd::CeilDiv(L.out_dim, 128)), dim3(128), 0, st,
                         L.Wiv, d_ivectors, L.out_dim, L.ivector_dim, nn->d_ivb, d_iv_rows);
      g.ivbias = nn->d_ivb; g.row2utt = row2utt;
    }
    if (L.bypass_layer != -2) {
      if (L.bypass_layer < 0) { g.byp = d_feats; g.ld_byp = ld_in; }
      else { g.byp = nn->act[L.bypass_layer]; g.ld_byp = nn->L[L.bypass_layer].out_pad; }
      g.bypmap = bypmap; g.bypass_scale = L.bypass_scale;
    }
    g.post_offset = L.log_softmax ? NULL : L.post_offset; g.post_scale = L.log_softmax ? 1.0f : L.post_scale;
    g.C = C + row_base * ldC; g.ldC = ldC;
    // tile choice: a narrow layer (bottleneck / prefinal, N <= 160) is ONE column tile as
    // wide as the layer, so the (large) A operand is streamed exactly once; wide layers use
    // the 128x128 tile.
    g.zeros = nn->d_zero; g.zeros_n = nn->d_zero; g.ones_n = nn->d_zero + nn->n_neutral;
    // epilogue: 2 = EpilogueWave (float4-aligned shapes), 3 = the same + the per-utterance i-vector bias, 1 = round 2's (any shape)
    const bool vec4 = (L.out_dim & 3) == 0 && (ldC & 3) == 0 && (!g.byp || (g.ld_byp & 3) == 0);
    static const bool epi1 = getenv("KAMD_GEMM_EPI1") != NULL && getenv("KAMD_GEMM_EPI1")[0] == '1';   // A/B against round 2's epilogue
    const int epi = (!vec4 || epi1) ? 1 : (g.ivbias ? 3 : 2);
    static const bool gen1 = getenv("KAMD_GEMM_GEN1") != NULL && getenv("KAMD_GEMM_GEN1")[0] == '1';   // A/B against the first generation
    const int nt32 = kamd::CeilDiv(L.out_dim, 32);
    // Kernel choice (round 3, measured with tools/microbench/gemm_lab on the layer shapes of the LibriSpeech topology):
    //   * the bottleneck layers with a long k loop (K >= 1024 into N = 129..160): persistent workgroups, two per CU;
    //   * layers wider than 160 whose form has a compiled epilogue: the fourth-generation kernel (static ring stages,
    //     scalar M0, epilogue compiled per layer form);
    //   * everything else: the second-generation one-tile kernels with the round-3 epilogue (EPI 2 / 3) or, for shapes
    //     that are not float4-aligned, round 2's (EPI 1).
    static const int gen_max = getenv("KAMD_GEMM_GEN") ? atoi(getenv("KAMD_GEMM_GEN")) : 4;          // A/B: 2 = second generation only
    const bool persist = gen_max >= 3 && epi == 2 && nt32 == 5 && L.in_pad / 16 >= 3 && L.n_off * L.in_pad >= 1024;
    const int ef = (g.bias ? kamd::EF_BIAS : 0) | (g.relu ? kamd::EF_RELU : 0) | (g.bn_scale ? kamd::EF_BN : 0) | (g.byp ? kamd::EF_BYP : 0) |
                   (g.post_offset ? kamd::EF_PO : 0) | (g.post_scale != 1.0f ? kamd::EF_SCALE : 0);
    bool gen4 = gen_max >= 4 && epi == 2 && nt32 > 5 && L.n_off * (L.in_pad / 16) >= 2;
    static const int n_cus = kamd_device_num_cus();
    static const bool no_lat = getenv("KAMD_GEMM_NO_LAT") != NULL;      // A/B: without the latency tiles of narrow layers over few rows
    if (!gen1 && gen4) {
      g.gx = kamd::CeilDiv(L.out_dim, 128); g.gy = static_cast<int>(kamd::CeilDiv(Ml, 128));
      dim3 grid(static_cast<unsigned>(g.gx) * static_cast<unsigned>(kamd::RoundUp(g.gy, 8)));
#define KAMD_G4(EFV) hipLaunchKernelGGL((kamd::TdnnGemmSaKernel<128, 128, 2, 2, EFV, false>), grid, dim3(256), 0, st, g)
      switch (ef) {
        case 0: KAMD_G4(0); break;
        case kamd::EF_BIAS: KAMD_G4(kamd::EF_BIAS); break;
        case kamd::EF_BN: KAMD_G4(kamd::EF_BN); break;
        case kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN: KAMD_G4(kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN); break;
        case kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN | kamd::EF_BYP: KAMD_G4(kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN | kamd::EF_BYP); break;
        case kamd::EF_BIAS | kamd::EF_PO: KAMD_G4(kamd::EF_BIAS | kamd::EF_PO); break;
        case kamd::EF_BIAS | kamd::EF_PO | kamd::EF_SCALE: KAMD_G4(kamd::EF_BIAS | kamd::EF_PO | kamd::EF_SCALE); break;
        default: gen4 = false; break;       // a layer form without a compiled epilogue
      }
#undef KAMD_G4
    }
    if (gen4 && !gen1) {
      // launched above
    } else if (!gen1 && gen_max >= 4 && !no_lat && epi == 2 && nt32 <= 5 && static_cast<int64_t>(kamd::CeilDiv(Ml, 128)) * nt32 <= 2 * n_cus) {
      // A narrow layer over few rows (a streaming tick: 24 rows per stream): one 128 x N workgroup per row tile walks the
      // whole k range alone and its time is k-blocks x DMA latency / blocks in flight (85 us for K = 1536 with a ring of
      // three, twelve such layers a tick).  Here, while all tiles fit on the chip at once (two workgroups per CU): 128 x 32
      // tiles, so that the N / 32 column tiles run on CUs of their own, and a ring of six stages (10 KB each), five
      // k-blocks in flight.  Same k order, same sums.
      dim3 grid(nt32, kamd::CeilDiv(Ml, 128));
      hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, 32, 4, 1, 6, 2>), grid, dim3(256), 0, st, g);
    } else if (persist && !gen1) {
      g.gy = static_cast<int>(kamd::CeilDiv(Ml, 128)); g.gx = 1;
      const int wgs = std::max(8, std::min(kamd::RoundUp(g.gy, 8), 2 * n_cus) & ~7);
      hipLaunchKernelGGL((kamd::TdnnGemmPersistKernel<128, 160, 4, 1, 3, false, 2, 1>), dim3(wgs), dim3(256), 0, st, g);
    } else if (!gen1) {
      if (nt32 <= 5) {
        dim3 grid(1, kamd::CeilDiv(Ml, 128));
        const bool e2 = epi == 2;       // (an i-vector layer this narrow keeps round 2's epilogue)
#define KAMD_TALL(BN) do { if (e2) hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, BN, 4, 1, 3, 2>), grid, dim3(256), 0, st, g); \
                           else hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, BN, 4, 1, 3, 1>), grid, dim3(256), 0, st, g); } while (0)
        switch (nt32) {
          case 1: KAMD_TALL(32); break;
          case 2: KAMD_TALL(64); break;
          case 3: KAMD_TALL(96); break;
          case 4: KAMD_TALL(128); break;
          default: KAMD_TALL(160); break;
        }
#undef KAMD_TALL
      } else {
        g.gx = kamd::CeilDiv(L.out_dim, 128); g.gy = static_cast<int>(kamd::CeilDiv(Ml, 128));
        dim3 grid(static_cast<unsigned>(g.gx) * static_cast<unsigned>(kamd::RoundUp(g.gy, 8)));
        if (epi == 2) hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, 128, 2, 2, 3, 2>), grid, dim3(256), 0, st, g);
        else if (epi == 3) hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, 128, 2, 2, 3, 3>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<128, 128, 2, 2, 3, 1>), grid, dim3(256), 0, st, g);
      }
    }
    if (!gen1) {
    } else if (nt32 <= 5) {
      dim3 grid(1, kamd::CeilDiv(Ml, 128));
      switch (nt32) {
        case 1: hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 32, 4, 1>), grid, dim3(256), 0, st, g); break;
        case 2: hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 64, 4, 1>), grid, dim3(256), 0, st, g); break;
        case 3: hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 96, 4, 1>), grid, dim3(256), 0, st, g); break;
        case 4: hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 128, 4, 1>), grid, dim3(256), 0, st, g); break;
        default: hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 160, 4, 1>), grid, dim3(256), 0, st, g); break;
      }
    } else {
      g.gx = kamd::CeilDiv(L.out_dim, 128); g.gy = static_cast<int>(kamd::CeilDiv(Ml, 128));
      dim3 grid(static_cast<unsigned>(g.gx) * static_cast<unsigned>(kamd::RoundUp(g.gy, 8)));
      hipLaunchKernelGGL((kamd::TdnnGemmKernel<128, 128, 2, 2>), grid, dim3(256), 0, st, g);
    }
    if (L.log_softmax)
      hipLaunchKernelGGL(kamd::LogSoftmaxRowsKernel, dim3(kamd::CeilDiv(Ml, 4)), dim3(256), 0, st, g.C, g.ldC,
                         static_cast<int>(Ml), L.out_dim, L.post_offset, L.post_scale);
    KAMD_HIP(hipGetLastError());
    flops += 2.0 * static_cast<double>(Ml) * L.out_dim * (L.n_off * L.in_dim + L.ivector_dim);
  }
  nn->last_flops = flops;
  return KAMD_OK;
}

int kamd_nnet_forward_batch_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_row_off,
                                   int ld_in, const float *d_ivectors, int n_utts, float *d_out,
                                   const int64_t *h_out_row_off, int ld_out, void *stream) {
  if (n_utts <= 0) return KAMD_OK;
  std::vector<int32_t> len(n_utts);
  for (int u = 0; u < n_utts; u++) len[u] = static_cast<int32_t>(h_in_row_off[u + 1] - h_in_row_off[u]);
  return ForwardItems(h, d_feats, h_in_row_off, len.data(), ld_in, d_ivectors, n_utts, d_out, h_out_row_off, ld_out, stream);
}

int kamd_nnet_forward_slices_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_start,
                                    const int32_t *h_in_len, int ld_in, const float *d_ivectors, int n_items,
                                    float *d_out, const int64_t *h_out_row_off, int ld_out, void *stream) {
  return ForwardItems(h, d_feats, h_in_start, h_in_len, ld_in, d_ivectors, n_items, d_out, h_out_row_off, ld_out, stream);
}

// The looped decodable's i-vector semantics (nnet3/nnet-compile-looped.cc:164-207 + ModifyNnetIvectorPeriod:
// the ivector input is read at Round(t, period)): the first layer's row at absolute time t takes row
// slot_base[i] + clamp(floor(t / period) - slot_first[i]) of the i-vector table; abs_t0[i] is the absolute
// time of item i's first input frame.  Everything else as kamd_nnet_forward_slices_device.
int kamd_nnet_forward_slices_slots_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_start, const int32_t *h_in_len,
                                          int ld_in, const float *d_ivector_table, int table_rows, int period,
                                          const int32_t *h_slot_base, const int32_t *h_slot_first, const int32_t *h_slot_count,
                                          const int32_t *h_abs_t0, int n_items, float *d_out, const int64_t *h_out_row_off,
                                          int ld_out, void *stream) {
  if (period <= 0 || table_rows <= 0) return kamd::SetError(KAMD_ERR_ARG, "bad i-vector slot table");
  SlotSpec sp = {period, table_rows, h_slot_base, h_slot_first, h_slot_count, h_abs_t0};
  return ForwardItems(h, d_feats, h_in_start, h_in_len, ld_in, d_ivector_table, n_items, d_out, h_out_row_off, ld_out, stream, &sp);
}

}  // extern "C"
namespace kamd {
// rows [src_row[i], +count[i]) of src -> rows [dst_row[i], ...) of dst, one workgroup row per copy item
__global__ void CopyRowBlocksKernel(const float *src, int ld_src, float *dst, int ld_dst, const int64_t *src_row,
                                    const int64_t *dst_row, const int *count, int cols) {
  const int item = blockIdx.y;
  for (int r = blockIdx.x; r < count[item]; r += gridDim.x)
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
      dst[(dst_row[item] + r) * ld_dst + c] = src[(src_row[item] + r) * ld_src + c];
}
int CopyRowBlocks(const float *src, int ld_src, float *dst, int ld_dst, const int64_t *d_src_row, const int64_t *d_dst_row,
                  const int *d_count, int n_items, int max_count, int cols, hipStream_t st) {
  if (n_items <= 0) return KAMD_OK;
  hipLaunchKernelGGL(CopyRowBlocksKernel, dim3(std::max(1, std::min(max_count, 64)), n_items), dim3(256), 0, st, src, ld_src, dst, ld_dst,
                     d_src_row, d_dst_row, d_count, cols);
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}
}  // namespace kamd
extern "C" {

// DecodableNnetSimple with online ivectors, for a batch of utterances (nnet3/nnet-am-decodable-
// simple.cc:93-214): every chunk of frames_per_chunk input frames (rounded up to a multiple of the subsampling
// factor, :278-310) is one item of a single batched forward, with its own left / right context
// (clamped at the utterance edges only) and the ivector row GetCurrentIvector picks for the middle
// of the chunk (:181-211).  The context rows are recomputed per chunk, as the reference does.
// rule 1 = NnetBatchComputer::SplitUtteranceIntoTasks (nnet3/nnet-batch-compute.cc:586-829; what nnet3-latgen-faster-batch
// evaluates): tasks of frames_per_chunk / subsampling output frames (integer division), the LAST task ending on the
// utterance's last frame and overlapping the one before it, the i-vector row of the task's middle
// ((begin_output_t + num_output_frames / 2) * f / period, the last row when at most 20 frames beyond the table).  A task
// is an item that computes exactly the output rows MergeTaskOutput (:832-870) keeps -- a feed-forward model's row does
// not depend on which other rows share its task, only on the task's i-vector.
static int ForwardChunkedRule(kamd_nnet *h, const float *d_feats, const int64_t *h_in_row_off, int ld_in,
                              const float *d_online_ivectors, const int64_t *h_iv_row_off, int iv_dim,
                              int ivector_period, int frames_per_chunk, int n_utts, float *d_out,
                              const int64_t *h_out_row_off, int ld_out, void *stream, int rule) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  if (n_utts <= 0) return KAMD_OK;
  const int sub = nn->subsampling;
  if (iv_dim != nn->L[0].ivector_dim || iv_dim <= 0) return kamd::SetError(KAMD_ERR_ARG, "model expects ivector dim %d, got %d", nn->L[0].ivector_dim, iv_dim);
  if (ivector_period <= 0 || frames_per_chunk <= 0) return kamd::SetError(KAMD_ERR_ARG, "bad ivector period / frames per chunk");
  if (rule == 0 && frames_per_chunk % sub != 0) frames_per_chunk = sub * ((frames_per_chunk + sub - 1) / sub);
  const int C = frames_per_chunk / sub;
  if (C <= 0) return kamd::SetError(KAMD_ERR_ARG, "frames per chunk %d below the subsampling factor %d", frames_per_chunk, sub);
  // Every chunk is an item whose layers are evaluated at exactly the times its outputs need (what the compiled
  // computation of a chunk contains); it writes its rows of d_out in place.  Neighbouring chunks that read the SAME
  // i-vector row are one item: their computations are identical on the rows they share (at the end of an utterance,
  // where GetCurrentIvector clamps to the last row, and whenever the period is not shorter than a chunk).
  static const bool no_merge = getenv("KAMD_CHUNK_NO_MERGE") != NULL && getenv("KAMD_CHUNK_NO_MERGE")[0] == '1';
  std::vector<int64_t> in_start, out_row;
  std::vector<int32_t> in_len, t0, nout, iv_row;
  const int64_t iv_base = h_iv_row_off[0];
  if (h_iv_row_off[n_utts] - iv_base > 2000000000LL) return kamd::SetError(KAMD_ERR_ARG, "online ivector table too large");
  for (int u = 0; u < n_utts; u++) {
    const int T = static_cast<int>(h_in_row_off[u + 1] - h_in_row_off[u]);
    const int n_iv = static_cast<int>(h_iv_row_off[u + 1] - h_iv_row_off[u]);
    if (T <= 0 || n_iv <= 0) return kamd::SetError(KAMD_ERR_ARG, "utterance %d has no frames / ivectors", u);
    const int n_out = (T + sub - 1) / sub;
    bool first = true;
    const int n_tasks = (n_out + C - 1) / C;
    for (int i = 0; i < n_tasks; i++) {
      const int start = i * C;                                  // first output row this item writes
      const int num = std::min(n_out - start, C);
      int ivf;
      if (rule == 0) {
        const int first_out = start * sub, last_out = (start + num - 1) * sub;
        ivf = (first_out + (last_out - first_out) / 2) / ivector_period;       // GetCurrentIvector (:181-211)
        if (ivf >= n_iv) {
          if ((ivf - (n_iv - 1)) * ivector_period > 50)
            return kamd::SetError(KAMD_ERR_ARG, "utterance %d: could not get iVector for frame %d (mismatched --online-ivector-period?)", u, first_out);
          ivf = n_iv - 1;
        }
      } else {
        // GetOutputFrameInfoForTasks: the task nominally covers C output frames from begin_output_t; the last of several
        // tasks is shifted back to end on the last frame (its first rows are the previous task's: not written again)
        const int begin_output_t = (n_tasks > 1 && i == n_tasks - 1) ? n_out - C : start;
        ivf = ((begin_output_t + C / 2) * sub) / ivector_period;               // AddOnlineIvectorsToTasks (:670-703)
        const int margin = (20 + ivector_period - 1) / ivector_period;
        if (ivf >= n_iv) {
          if (ivf > n_iv - margin) ivf = n_iv - 1;
          else return kamd::SetError(KAMD_ERR_ARG, "utterance %d: could not get iVector for frame %d, online-ivectors matrix has %d rows (mismatched --online-ivector-period?)", u, ivf, n_iv);
        }
      }
      const int32_t row = static_cast<int32_t>(h_iv_row_off[u] - iv_base + ivf);
      if (!first && !no_merge && iv_row.back() == row) { nout.back() += num; continue; }
      first = false;
      in_start.push_back(h_in_row_off[u]); in_len.push_back(T);
      t0.push_back(start * sub); nout.push_back(num);
      out_row.push_back(h_out_row_off[u] + start);
      iv_row.push_back(row);
    }
  }
  const int n_items = static_cast<int>(in_start.size());
  out_row.push_back(h_out_row_off[n_utts - 1] + (h_in_row_off[n_utts] - h_in_row_off[n_utts - 1] + sub - 1) / sub);
  ChunkSpec cs = {t0.data(), nout.data(), iv_row.data()};
  return ForwardItems(h, d_feats, in_start.data(), in_len.data(), ld_in, d_online_ivectors + iv_base * iv_dim, n_items, d_out, out_row.data(),
                      ld_out, stream, NULL, &cs);
}

int kamd_nnet_forward_chunked_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_row_off, int ld_in,
                                     const float *d_online_ivectors, const int64_t *h_iv_row_off, int iv_dim,
                                     int ivector_period, int frames_per_chunk, int n_utts, float *d_out,
                                     const int64_t *h_out_row_off, int ld_out, void *stream) {
  return ForwardChunkedRule(h, d_feats, h_in_row_off, ld_in, d_online_ivectors, h_iv_row_off, iv_dim, ivector_period, frames_per_chunk, n_utts,
                            d_out, h_out_row_off, ld_out, stream, 0);
}

int kamd_nnet_forward_tasks_device(kamd_nnet *h, const float *d_feats, const int64_t *h_in_row_off, int ld_in,
                                   const float *d_online_ivectors, const int64_t *h_iv_row_off, int iv_dim,
                                   int ivector_period, int frames_per_chunk, int n_utts, float *d_out,
                                   const int64_t *h_out_row_off, int ld_out, void *stream) {
  return ForwardChunkedRule(h, d_feats, h_in_row_off, ld_in, d_online_ivectors, h_iv_row_off, iv_dim, ivector_period, frames_per_chunk, n_utts,
                            d_out, h_out_row_off, ld_out, stream, 1);
}

// ---- one minibatch of NnetInferenceTasks (nnet3/nnet-batch-compute.h:42-110), the unit NnetBatchComputer::Compute evaluates
// (nnet-batch-compute.cc:398-470 FormatInputs / computer.Run / FormatOutputs): the entry a Kaldi-side NnetBatchComputer binds
// when it keeps its own scheduler (AcceptTask, priorities, full / partial minibatches) and hands the device the tasks it
// picked.  Task i = output frames [first_output_t, + num_output_frames) (in units of the subsampled rate) of the utterance
// whose features are rows [in_row, + in_len) of d_feats -- the context beyond the utterance's ends is the first / last frame
// repeated, as SplitInputToTasks pads it (:705-770) -- with row iv_row of d_ivectors (an online i-vector or the utterance's
// own; -1 for a model without the input).  Outputs: the tasks back to back in d_out, num_output_frames rows each.
int kamd_nnet_forward_inference_tasks_device(kamd_nnet *h, const float *d_feats, int ld_in, const float *d_ivectors, int iv_dim,
                                             const kamd_inference_task *tasks, int n_tasks, float *d_out, int ld_out, void *stream) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  if (n_tasks <= 0) return KAMD_OK;
  if (!tasks || !d_feats || !d_out) return kamd::SetError(KAMD_ERR_ARG, "inference tasks: null argument");
  const int sub = nn->subsampling, want_iv = nn->L[0].ivector_dim;
  if (want_iv > 0 && (!d_ivectors || iv_dim != want_iv)) return kamd::SetError(KAMD_ERR_ARG, "model expects ivector dim %d, got %d", want_iv, d_ivectors ? iv_dim : 0);
  std::vector<int64_t> in_start(n_tasks), out_row(n_tasks + 1, 0);
  std::vector<int32_t> in_len(n_tasks), t0(n_tasks), nout(n_tasks), iv_row(n_tasks);
  for (int i = 0; i < n_tasks; i++) {
    const kamd_inference_task &t = tasks[i];
    if (t.in_len <= 0 || t.first_output_t < 0 || t.num_output_frames <= 0 || t.in_row < 0)
      return kamd::SetError(KAMD_ERR_ARG, "inference task %d: bad range", i);
    if (want_iv > 0 && t.iv_row < 0) return kamd::SetError(KAMD_ERR_ARG, "inference task %d: the model needs an i-vector row", i);
    in_start[i] = t.in_row; in_len[i] = t.in_len; t0[i] = t.first_output_t * sub; nout[i] = t.num_output_frames;
    iv_row[i] = want_iv > 0 ? t.iv_row : 0;
    out_row[i + 1] = out_row[i] + t.num_output_frames;
  }
  ChunkSpec cs = {t0.data(), nout.data(), iv_row.data()};
  return ForwardItems(h, d_feats, in_start.data(), in_len.data(), ld_in, want_iv > 0 ? d_ivectors : NULL, n_tasks, d_out, out_row.data(), ld_out,
                      stream, NULL, &cs);
}

// ---- nnet3::Component::Propagate (nnet3/nnet-component-itf.h:130-132) for ONE fused layer: SURVEY 8(b)'s option (i),
// the compatibility entry for a host that keeps nnet3's own computation and hands single components to the device.
// TdnnComponent::Propagate's shape (nnet-tdnn-component.cc:181-212): `in` holds consecutive time steps, out row j is
// sum_i W_i in[j + off_i - off_min] (+ bias, ReLU, BatchNorm of the fused layer), so out has in_rows - (off_max - off_min)
// rows.  Built on the batched forward: the layer as a one-layer network over the in_rows "frames", whose interior rows
// (the ones no edge clamp touches) are exactly those outputs.
struct Component {
  kamd_nnet *net; int in_dim, out_dim, off_min, off_max, ld;
  float *d_in = NULL, *d_out = NULL; size_t in_cap = 0, out_cap = 0;
};

kamd_component *kamd_component_create(const kamd_layer_desc *layer) {
  if (!layer || layer->ivector_dim != 0 || layer->n_offsets < 1 || layer->multi_input) { kamd::SetError(KAMD_ERR_ARG, "kamd_component_create: a plain fused layer is expected"); return NULL; }
  kamd_layer_desc d = *layer;
  d.input_layer = -1; d.bypass_layer = -2; d.bypass_scale = 0.0f;      // (a bypass is a Sum descriptor of the graph, not of the component)
  kamd_nnet *net = kamd_nnet_create(&d, 1, d.in_dim, 1);
  if (!net) return NULL;
  Component *c = new Component();
  c->net = net; c->in_dim = d.in_dim; c->out_dim = d.out_dim; c->ld = kamd::RoundUp(d.in_dim, 16);
  c->off_min = d.offsets[0]; c->off_max = d.offsets[0];
  for (int i = 1; i < d.n_offsets; i++) { c->off_min = std::min(c->off_min, d.offsets[i]); c->off_max = std::max(c->off_max, d.offsets[i]); }
  if (c->off_min > 0 || c->off_max < 0) {   // (every recipe's offsets straddle 0; the interior-row argument below needs it)
    kamd::SetError(KAMD_ERR_ARG, "kamd_component_create: time offsets must include values <= 0 and >= 0");
    kamd_nnet_destroy(net); delete c; return NULL;
  }
  return reinterpret_cast<kamd_component *>(c);
}
void kamd_component_destroy(kamd_component *h) {
  Component *c = reinterpret_cast<Component *>(h);
  if (!c) return;
  kamd_nnet_destroy(c->net);
  if (c->d_in) (void)hipFree(c->d_in);
  if (c->d_out) (void)hipFree(c->d_out);
  delete c;
}
int kamd_component_output_rows(const kamd_component *h, int in_rows) {
  const Component *c = reinterpret_cast<const Component *>(h);
  return std::max(0, in_rows - (c->off_max - c->off_min));
}
int kamd_component_propagate(kamd_component *h, const float *d_in, int in_rows, int ld_in, float *d_out, int ld_out, void *stream) {
  Component *c = reinterpret_cast<Component *>(h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n_out = kamd_component_output_rows(h, in_rows);
  if (n_out <= 0) return 0;
  if (ld_in < c->in_dim || ld_out < c->out_dim) return kamd::SetError(KAMD_ERR_ARG, "kamd_component_propagate: leading dimension smaller than the matrix");
  // staging: the GEMM reads rows of 16-float multiples whose pad columns are zero
  const size_t need_in = static_cast<size_t>(in_rows) * c->ld, need_out = static_cast<size_t>(in_rows) * c->out_dim;
  if (need_in > c->in_cap) { if (c->d_in) (void)hipFree(c->d_in); c->d_in = NULL; KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_in), need_in * 4)); c->in_cap = need_in; }
  if (need_out > c->out_cap) { if (c->d_out) (void)hipFree(c->d_out); c->d_out = NULL; KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_out), need_out * 4)); c->out_cap = need_out; }
  KAMD_HIP(hipMemsetAsync(c->d_in, 0, need_in * 4, st));
  KAMD_HIP(hipMemcpy2DAsync(c->d_in, static_cast<size_t>(c->ld) * 4, d_in, static_cast<size_t>(ld_in) * 4, static_cast<size_t>(c->in_dim) * 4, in_rows,
                            hipMemcpyDeviceToDevice, st));
  int64_t in_off[2] = {0, in_rows}, out_off[1] = {0};
  const int rc = kamd_nnet_forward_batch_device(c->net, c->d_in, in_off, c->ld, NULL, 1, c->d_out, out_off, c->out_dim, st);
  if (rc != KAMD_OK) return rc;
  // forward row t uses in[clamp(t + off_i)]: rows t in [-off_min, in_rows - off_max) are clamp-free = Propagate's rows
  KAMD_HIP(hipMemcpy2DAsync(d_out, static_cast<size_t>(ld_out) * 4, c->d_out + static_cast<size_t>(-c->off_min) * c->out_dim,
                            static_cast<size_t>(c->out_dim) * 4, static_cast<size_t>(c->out_dim) * 4, n_out, hipMemcpyDeviceToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  return n_out;
}

int kamd_nnet_forward(kamd_nnet *h, const float *feats, int T, const float *ivector, float *out,
                      int out_rows_cap) {
  Nnet *nn = reinterpret_cast<Nnet *>(h);
  if (T <= 0) return 0;
  const int n_out = (T + nn->subsampling - 1) / nn->subsampling, P = nn->L.back().out_dim;
  if (n_out > out_rows_cap) return kamd::SetError(KAMD_ERR_ARG, "output buffer too small");
  const int ld = nn->L[0].in_pad, ivd = nn->L[0].ivector_dim;
  std::vector<float> padded(static_cast<size_t>(T) * ld, 0.0f);
  for (int t = 0; t < T; t++) memcpy(&padded[static_cast<size_t>(t) * ld], feats + static_cast<size_t>(t) * nn->input_dim, sizeof(float) * nn->input_dim);
  float *d_in = NULL, *d_out = NULL, *d_iv = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_in), padded.size() * sizeof(float)));
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_out), static_cast<size_t>(n_out) * P * sizeof(float)));
  KAMD_HIP(hipMemcpy(d_in, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice));
  if (ivd > 0) {
    if (!ivector) return kamd::SetError(KAMD_ERR_ARG, "model needs an ivector");
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_iv), ivd * sizeof(float)));
    KAMD_HIP(hipMemcpy(d_iv, ivector, ivd * sizeof(float), hipMemcpyHostToDevice));
  }
  int64_t in_off[2] = {0, T}, out_off[1] = {0};
  int rc = kamd_nnet_forward_batch_device(h, d_in, in_off, ld, d_iv, 1, d_out, out_off, P, NULL);
  if (rc == KAMD_OK) {
    hipError_t e = hipMemcpy(out, d_out, static_cast<size_t>(n_out) * P * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "D2H failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_in); (void)hipFree(d_out);
  if (d_iv) (void)hipFree(d_iv);
  return rc == KAMD_OK ? n_out : rc;
}

}  // extern "C"
