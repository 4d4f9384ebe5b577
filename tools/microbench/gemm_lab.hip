// gemm_lab -- the acoustic model's GEMM kernels (kaldi_amd/csrc/nnet.hip, compiled in as they are) on the layer shapes of
// the LibriSpeech TDNN-F topology, one launch at a time: TFLOP/s per variant, bit-equality of every variant's output
// with the first one, and where a workgroup's time goes (s_memtime stamps: row-map prologue, ring fill, first k-block's
// wait, main loop, epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DKAMD_GEMM_LAB -I include -I kaldi_amd/csrc \
//         tools/microbench/gemm_lab.hip kaldi_amd/csrc/common.cc -o tools/microbench/gemm_lab
//   tools/microbench/gemm_lab [rows]
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../kaldi_amd/csrc/nnet.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using kamd::GemmArgs;

struct Shape { const char *name; int in_dim, n_off, offs[2], N; bool epi; bool bypass; };

static float *DevRand(size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (static_cast<float>((s >> 8) & 0xFFFF) - 32768.0f) * (scale / 32768.0f); }
  float *d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

struct Variant { std::string name; std::function<void(GemmArgs &, int M, int N, hipStream_t)> launch; int prio = 0; int aux = 0; int valu = 0; };

template <int BM, int BN, int WM, int WN, int NST, int EPI>
static void LaunchTall(GemmArgs &g, int M, int N, hipStream_t st) {      // one column tile as wide as the layer
  g.gx = 0; g.gy = 0;
  hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<BM, BN, WM, WN, NST, EPI>), dim3(1, kamd::CeilDiv(M, BM)), dim3(256), 0, st, g);
}
template <int BM, int BN, int WM, int WN, int NST, int EPI>
static void LaunchGrid(GemmArgs &g, int M, int N, hipStream_t st) {      // XCD-aware 1-D order over a 2-D tile grid
  g.gx = kamd::CeilDiv(N, BN); g.gy = kamd::CeilDiv(M, BM);
  hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<BM, BN, WM, WN, NST, EPI>), dim3(static_cast<unsigned>(g.gx) * kamd::RoundUp(g.gy, 8)), dim3(256), 0, st, g);
}

template <int BM, int BN, int WM, int WN, int NST, int BK, int OCC>
static void LaunchTallK(GemmArgs &g, int M, int N, hipStream_t st) {
  g.gx = 0; g.gy = 0;
  hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<BM, BN, WM, WN, NST, 2, BK, OCC>), dim3(1, kamd::CeilDiv(M, BM)), dim3(256), 0, st, g);
}
template <int BM, int BN, int WM, int WN, int NST, int BK, int OCC>
static void LaunchGridK(GemmArgs &g, int M, int N, hipStream_t st) {
  g.gx = kamd::CeilDiv(N, BN); g.gy = kamd::CeilDiv(M, BM);
  hipLaunchKernelGGL((kamd::TdnnGemmDmaKernel<BM, BN, WM, WN, NST, 2, BK, OCC>), dim3(static_cast<unsigned>(g.gx) * kamd::RoundUp(g.gy, 8)), dim3(256), 0, st, g);
}
template <int BM, int BN, int WM, int WN, int EF, int NST = 3, int OCC = 3>
static void LaunchSa(GemmArgs &g, int M, int N, hipStream_t st) {
  if (N <= BN) { g.gx = 0; g.gy = 0; hipLaunchKernelGGL((kamd::TdnnGemmSaKernel<BM, BN, WM, WN, EF, false, NST, OCC>), dim3(1, kamd::CeilDiv(M, BM)), dim3(256), 0, st, g); return; }
  g.gx = kamd::CeilDiv(N, BN); g.gy = kamd::CeilDiv(M, BM);
  hipLaunchKernelGGL((kamd::TdnnGemmSaKernel<BM, BN, WM, WN, EF, false, NST, OCC>), dim3(static_cast<unsigned>(g.gx) * kamd::RoundUp(g.gy, 8)), dim3(256), 0, st, g);
}
template <int BM, int BN, int WM, int WN, int NST, int OCC>
static void LaunchLoader(GemmArgs &g, int M, int N, hipStream_t st) {
  if (N <= BN) { g.gx = 0; g.gy = 0; hipLaunchKernelGGL((kamd::TdnnGemmLoaderKernel<BM, BN, WM, WN, NST, OCC, false>), dim3(1, kamd::CeilDiv(M, BM)), dim3(320), 0, st, g); return; }
  g.gx = kamd::CeilDiv(N, BN); g.gy = kamd::CeilDiv(M, BM);
  hipLaunchKernelGGL((kamd::TdnnGemmLoaderKernel<BM, BN, WM, WN, NST, OCC, false>), dim3(static_cast<unsigned>(g.gx) * kamd::RoundUp(g.gy, 8)), dim3(320), 0, st, g);
}
template <int BM, int BN, int WM, int WN, int NST, int OCC, int DEPTH>
static void LaunchPersist(GemmArgs &g, int M, int N, hipStream_t st) {   // persistent workgroups, OCC per compute unit
  g.gx = kamd::CeilDiv(N, BN); g.gy = kamd::CeilDiv(M, BM);
  int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int wgs = std::max(8, std::min(g.gx * kamd::RoundUp(g.gy, 8), OCC * cus) & ~7);
  hipLaunchKernelGGL((kamd::TdnnGemmPersistKernel<BM, BN, WM, WN, NST, false, OCC, DEPTH>), dim3(wgs), dim3(256), 0, st, g);
}

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 333000;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const Shape shapes[] = {{"affine 2x160 -> 1536 (+bias relu bn bypass)", 160, 2, {0, 1}, 1536, true, true},
                          {"linear 2x1536 -> 160", 1536, 2, {-1, 0}, 160, false, false},
                          {"output 256 -> 6000 (+bias, post offset)", 256, 1, {0, 0}, 6000, true, false},
                          {"long-K probe 2x1536 -> 1536 (main loop only matters)", 1536, 2, {-1, 0}, 1536, false, false}};
  const int only = argc > 3 ? atoi(argv[3]) : -1;
  int shape_no = -1;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *zeros;
  { std::vector<float> z(16384, 0.0f); std::fill(z.begin() + 8192, z.end(), 1.0f); CK(hipMalloc(&zeros, z.size() * 4)); CK(hipMemcpy(zeros, z.data(), z.size() * 4, hipMemcpyHostToDevice)); }
  for (const Shape &sh : shapes) {
    if (only >= 0 && ++shape_no != only) continue;
    const int in_pad = kamd::RoundUp(sh.in_dim, 16), K = sh.n_off * in_pad, N = sh.N, N_pad = kamd::RoundUp(N, 128);
    float *A = DevRand(static_cast<size_t>(M) * in_pad, 1, 1.0f);
    float *W = DevRand(static_cast<size_t>(N_pad) * K, 2, 0.05f);
    float *bias = DevRand(N, 3, 0.1f), *bs = DevRand(N, 4, 1.0f), *bo = DevRand(N, 5, 0.1f), *po = DevRand(N, 6, 1.0f);
    float *byp = sh.bypass ? DevRand(static_cast<size_t>(M) * N, 7, 1.0f) : NULL;
    std::vector<int> rmh(static_cast<size_t>(sh.n_off + 1) * M);
    for (int o = 0; o < sh.n_off; o++)
      for (int m = 0; m < M; m++) { int r = m + sh.offs[o]; rmh[static_cast<size_t>(o) * M + m] = r < 0 ? 0 : (r >= M ? M - 1 : r); }
    for (int m = 0; m < M; m++) rmh[static_cast<size_t>(sh.n_off) * M + m] = m;
    int *rowmap; CK(hipMalloc(&rowmap, rmh.size() * 4)); CK(hipMemcpy(rowmap, rmh.data(), rmh.size() * 4, hipMemcpyHostToDevice));
    float *C[2]; for (int i = 0; i < 2; i++) { CK(hipMalloc(&C[i], static_cast<size_t>(M) * N * 4)); CK(hipMemset(C[i], 0, static_cast<size_t>(M) * N * 4)); }
    const int wgs_max = 8 * (kamd::CeilDiv(M, 64) + 8) * kamd::CeilDiv(N, 32);
    unsigned long long *stamps; CK(hipMalloc(&stamps, static_cast<size_t>(wgs_max) * 256));
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.A = A; g.ldA = in_pad; g.rowmap = rowmap; g.M = M; g.N = N; g.n_off = sh.n_off; g.in_pad = in_pad; g.W = W;
    if (sh.epi) { g.bias = bias; g.post_offset = po; }
    if (sh.bypass) { g.relu = 1; g.bn_scale = bs; g.bn_offset = bo; g.byp = byp; g.ld_byp = N; g.bypmap = rowmap + static_cast<size_t>(sh.n_off) * M; g.bypass_scale = 0.75f; g.post_offset = NULL; }
    g.post_scale = 1.0f; g.ldC = N; g.zeros = zeros; g.zeros_n = zeros; g.ones_n = zeros + 8192;
    std::vector<Variant> vs;
    if (N <= 160) {
      vs.push_back({"128x160 4x1 ring3 epi1 (round 2)", LaunchTall<128, 160, 4, 1, 3, 1>});
      vs.push_back({"128x160 4x1 ring3 epi2", LaunchTall<128, 160, 4, 1, 3, 2>});
      vs.push_back({"128x160 4x1 ring3 persistent x2 d1", LaunchPersist<128, 160, 4, 1, 3, 2, 1>});
      vs.push_back({"128x160 4x1 gen4 (scalar addressing, specialised epilogue)", LaunchSa<128, 160, 4, 1, 0>});
      vs.push_back({"128x160 4x1 gen4 ring2 x3", LaunchSa<128, 160, 4, 1, 0, 2, 3>});
    } else {
      vs.push_back({"128x128 2x2 ring3 epi1 (round 2)", LaunchGrid<128, 128, 2, 2, 3, 1>});
      vs.push_back({"128x128 2x2 ring3 epi2", LaunchGrid<128, 128, 2, 2, 3, 2>});
      vs.push_back({"128x128 2x2 ring3 persistent x3 d1", LaunchPersist<128, 128, 2, 2, 3, 3, 1>});
      if (sh.bypass) vs.push_back({"128x128 2x2 gen4 (scalar addressing, specialised epilogue)", LaunchSa<128, 128, 2, 2, kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN | kamd::EF_BYP>});
      else if (sh.epi) vs.push_back({"128x128 2x2 gen4 (scalar addressing, specialised epilogue)", LaunchSa<128, 128, 2, 2, kamd::EF_BIAS | kamd::EF_PO>});
      else vs.push_back({"128x128 2x2 gen4 (scalar addressing, specialised epilogue)", LaunchSa<128, 128, 2, 2, 0>});
      if (sh.bypass) vs.push_back({"128x128 2x2 gen4 ring2 x4", LaunchSa<128, 128, 2, 2, kamd::EF_BIAS | kamd::EF_RELU | kamd::EF_BN | kamd::EF_BYP, 2, 4>});
      else if (sh.epi) vs.push_back({"128x128 2x2 gen4 ring2 x4", LaunchSa<128, 128, 2, 2, kamd::EF_BIAS | kamd::EF_PO, 2, 4>});
      else vs.push_back({"128x128 2x2 gen4 ring2 x4", LaunchSa<128, 128, 2, 2, 0, 2, 4>});
    }
    const double flops = 2.0 * M * static_cast<double>(N) * sh.n_off * sh.in_dim;
    printf("== %s, %d rows\n", sh.name, M);
    for (size_t v = 0; v < vs.size(); v++) {
      g.C = C[v == 0 ? 0 : 1]; g.stamps = NULL; g.lab_prio = vs[v].prio; g.lab_aux = vs[v].aux; g.lab_valu = vs[v].valu;
      for (int r = 0; r < 3; r++) vs[v].launch(g, M, N, st);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; r++) vs[v].launch(g, M, N, st);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      // stamps
      CK(hipMemset(stamps, 0, static_cast<size_t>(wgs_max) * 256));
      g.stamps = stamps; vs[v].launch(g, M, N, st); CK(hipStreamSynchronize(st)); g.stamps = NULL;
      std::vector<unsigned long long> hs(static_cast<size_t>(wgs_max) * 32);
      CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
      double seg[5] = {0, 0, 0, 0, 0}, clk = 0; size_t cnt = 0;
      for (int w = 0; w < wgs_max; w++) {
        const unsigned long long *q = &hs[static_cast<size_t>(w) * 32];
        if (!q[0] || !q[5]) continue;
        if (q[1]) for (int i = 0; i < 5; i++) seg[i] += static_cast<double>(q[i + 1] - q[i]);
        clk += static_cast<double>(q[5] - q[0]) / static_cast<double>(q[7] - q[6]) * 0.1;     // s_memrealtime: 100 MHz -> GHz
        cnt++;
      }
      clk /= cnt ? cnt : 1;
      // a second stamped launch that also times the segments of every k-block, per wave (perturbs the loop: shares, not totals)
      double lseg[4] = {0, 0, 0, 0};
      if (vs[v].name.find("persistent") == std::string::npos && vs[v].name.find("loader") == std::string::npos && vs[v].name.find("gen4") == std::string::npos) {
        CK(hipMemset(stamps, 0, static_cast<size_t>(wgs_max) * 256));
        g.stamps = stamps; g.lab_loop = 1; vs[v].launch(g, M, N, st); CK(hipStreamSynchronize(st)); g.stamps = NULL; g.lab_loop = 0;
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        size_t c2 = 0;
        for (int w = 0; w < wgs_max; w++) {
          const unsigned long long *q = &hs[static_cast<size_t>(w) * 32];
          if (!q[0] || !q[5]) continue;
          for (int wv = 0; wv < 4; wv++) for (int i = 0; i < 4; i++) lseg[i] += static_cast<double>(q[8 + wv * 4 + i]) / 4;
          c2++;
        }
        const int nkb = K / (vs[v].name.find("BK32") != std::string::npos ? 32 : 16);
        for (int i = 0; i < 4; i++) lseg[i] /= (c2 ? c2 : 1) * static_cast<double>(nkb);
      }
      for (int i = 0; i < 5; i++) seg[i] /= cnt ? cnt : 1;
      size_t bad = 0;
      if (v > 0) {
        std::vector<float> a(static_cast<size_t>(M) * N), b(a.size());
        CK(hipMemcpy(a.data(), C[0], a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), C[1], b.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); i++) bad += memcmp(&a[i], &b[i], 4) != 0;
        CK(hipMemset(C[1], 0, a.size() * 4));
      }
      printf("  %-36s %7.3f ms %6.1f TFLOP/s (%.3f of 157.3) clock %.2f GHz | wg cycles: rowmap %.0f ringfill %.0f first-wait %.0f mainloop %.0f epilogue %.0f (%zu wgs) | per k-block: dma-wait %.0f barrier %.0f issue %.0f lds+mfma %.0f%s\n",
             vs[v].name.c_str(), ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, clk, seg[0], seg[1], seg[2], seg[3], seg[4], cnt, lseg[0], lseg[1], lseg[2], lseg[3],
             v == 0 ? "" : (bad ? "  OUTPUT DIFFERS" : "  bit-equal"));
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(bs)); CK(hipFree(bo)); CK(hipFree(po)); if (byp) CK(hipFree(byp));
    CK(hipFree(rowmap)); CK(hipFree(C[0])); CK(hipFree(C[1])); CK(hipFree(stamps));
  }
  return 0;
}
