// oracle/orc_nnet_blas.cc -- TEST / BASELINE INFRASTRUCTURE ONLY (CPU; never shipped).
// Only tests/ and bench.py's cpu_baseline leg load it.
//
// The nnet3 forward the way the REFERENCE runs it on a CPU, for timing the CPU baseline with the reference's own
// arithmetic path instead of orc_nnet.cc's scalar dot products:
//   * DecodableNnetSimple evaluates the utterance chunk by chunk (nnet3/nnet-am-decodable-simple.cc:93-167):
//     frames_per_chunk input frames rounded up to a multiple of the frame-subsampling factor (:278-310), every chunk with
//     its own left / right context (first / last frame repeated at the utterance edges, :147-160), context rows
//     recomputed per chunk;
//   * inside a chunk every node is evaluated only at the time indexes its consumers request (the compiled
//     computation, nnet3/nnet-compile.cc), and every component's Propagate is BLAS sgemm over all those rows:
//     AffineComponent nnet-simple-component.cc:1234-1243, TdnnComponent one AddMatMat per time offset
//     (nnet-tdnn-component.cc:201-210), CuMatrix::AddMatMat -> cblas_sgemm on the CPU (matrix/kaldi-matrix.cc:182,
//     matrix/cblas-wrappers.h:233).
// The sgemm itself is not linked: the caller passes cblas_sgemm's address (bench.py / the tests take it from the
// OpenBLAS that numpy ships, ILP64 interface, one BLAS thread per caller thread: nnet3-latgen-faster is
// single-threaded per job).  Same fused-layer model as orc_nnet.cc; results equal to fp32 rounding
// (tests/test_oracle_nnet.py).  oracle/orc_blas.py is the same restatement in numpy.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/kaldi_amd.h"

namespace {

// cblas_sgemm with 64-bit integers (OpenBLAS ILP64): layout 101 = row major, trans 111 = no, 112 = yes
typedef void (*SgemmFn)(int layout, int transa, int transb, int64_t m, int64_t n, int64_t k, float alpha, const float *a, int64_t lda,
                        const float *b, int64_t ldb, float beta, float *c, int64_t ldc);

struct TimeSet {            // sorted time indexes a node is evaluated at (in one chunk)
  std::vector<int> t;
  int Find(int x) const { return static_cast<int>(std::lower_bound(t.begin(), t.end(), x) - t.begin()); }
  void Add(const std::vector<int> &more) {
    std::vector<int> u(t.size() + more.size());
    u.resize(std::set_union(t.begin(), t.end(), more.begin(), more.end(), u.begin()) - u.begin());
    t.swap(u);
  }
};

}  // namespace

// online_ivectors != NULL: --online-ivectors of the recipe (steps/nnet3/decode.sh:105-107) -- every chunk is evaluated with the row
// GetCurrentIvector (nnet-am-decodable-simple.cc:181-211) picks for its middle, as orc_nnet.cc's orc_nnet_forward_chunked does;
// `ivector` is then ignored.  Returns -2 when a chunk has no i-vector ("Could not get iVector for frame").
static int ForwardBlas(const kamd_layer_desc *L, int n_layers, int input_dim, int subsampling, const float *feats, int T, const float *ivector,
                       const float *online_ivectors, int n_iv_rows, int iv_dim, int ivector_period, int frames_per_chunk, void *sgemm_ptr,
                       float *out, int out_rows_cap) {
  SgemmFn sgemm = reinterpret_cast<SgemmFn>(sgemm_ptr);
  if (T <= 0 || !sgemm) return -1;
  for (int l = 0; l < n_layers; l++) if (L[l].multi_input) return -3;     // single-producer layers only (the bench models); orc_nnet_forward does the rest
  const int sub = subsampling, n_out = (T + sub - 1) / sub, P = L[n_layers - 1].out_dim;
  if (n_out > out_rows_cap) return -1;
  const int C = frames_per_chunk <= 0 ? n_out : (frames_per_chunk + sub - 1) / sub;
  std::vector<TimeSet> req(n_layers + 1);                 // index n_layers = the network input
  std::vector<std::vector<float> > act(n_layers + 1);
  std::vector<float> gathered;
  for (int start = 0; start < n_out; start += C) {
    const int num = std::min(C, n_out - start);
    if (online_ivectors) {
      const int first_output_frame = start * sub, last_output_frame = (start + num - 1) * sub;
      const int frame_to_search = first_output_frame + (last_output_frame - first_output_frame) / 2;
      int ivector_frame = frame_to_search / ivector_period;
      if (ivector_frame >= n_iv_rows) {
        if ((ivector_frame - (n_iv_rows - 1)) * ivector_period > 50) return -2;
        ivector_frame = n_iv_rows - 1;
      }
      ivector = online_ivectors + static_cast<int64_t>(ivector_frame) * iv_dim;
    }
    for (TimeSet &r : req) r.t.clear();
    req[n_layers - 1].t.resize(num);
    for (int i = 0; i < num; i++) req[n_layers - 1].t[i] = (start + i) * sub;
    for (int l = n_layers - 1; l >= 0; l--) {             // what every node must provide (ComputationRequest -> indexes)
      const kamd_layer_desc &d = L[l];
      const std::vector<int> &t = req[l].t;
      const int src = d.input_layer < 0 ? n_layers : d.input_layer;
      for (int o = 0; o < d.n_offsets; o++) {
        std::vector<int> need(t);
        for (int &x : need) x += d.offsets[o];
        req[src].Add(need);
      }
      if (d.bypass_layer != -2) req[d.bypass_layer < 0 ? n_layers : d.bypass_layer].Add(t);
    }
    {                                                     // input rows, edge frames repeated
      const std::vector<int> &t = req[n_layers].t;
      act[n_layers].resize(t.size() * static_cast<size_t>(input_dim));
      for (size_t i = 0; i < t.size(); i++) {
        const int c = std::min(std::max(t[i], 0), T - 1);
        memcpy(&act[n_layers][i * input_dim], feats + static_cast<size_t>(c) * input_dim, sizeof(float) * input_dim);
      }
    }
    for (int l = 0; l < n_layers; l++) {
      const kamd_layer_desc &d = L[l];
      const std::vector<int> &t = req[l].t;
      const int rows = static_cast<int>(t.size()), N = d.out_dim, K = d.n_offsets * d.in_dim + d.ivector_dim;
      const int src = d.input_layer < 0 ? n_layers : d.input_layer;
      std::vector<float> &y = act[l];
      y.resize(static_cast<size_t>(rows) * N);
      for (int r = 0; r < rows; r++)                      // bias first, then AddMatMat with beta = 1 (nnet-tdnn-component.cc:189-210)
        for (int n = 0; n < N; n++) y[static_cast<size_t>(r) * N + n] = d.bias ? d.bias[n] : 0.0f;
      for (int o = 0; o < d.n_offsets; o++) {
        // the operand rows of this offset (kCopyRows / a strided view in the reference), then one sgemm:
        // y[rows x N] += x[rows x in_dim] * W_o[N x in_dim]^T, W_o = columns [o * in_dim, +in_dim) of W (ld = K)
        gathered.resize(static_cast<size_t>(rows) * d.in_dim);
        for (int r = 0; r < rows; r++)
          memcpy(&gathered[static_cast<size_t>(r) * d.in_dim], &act[src][static_cast<size_t>(req[src].Find(t[r] + d.offsets[o])) * d.in_dim],
                 sizeof(float) * d.in_dim);
        sgemm(101, 111, 112, rows, N, d.in_dim, 1.0f, gathered.data(), d.in_dim, d.W + static_cast<size_t>(o) * d.in_dim, K, 1.0f, y.data(), N);
      }
      if (d.ivector_dim > 0) {                            // ReplaceIndex(ivector, t, 0): the same vector on every row
        std::vector<float> ivb(N, 0.0f);
        sgemm(101, 111, 112, 1, N, d.ivector_dim, 1.0f, ivector, d.ivector_dim, d.W + static_cast<size_t>(d.n_offsets) * d.in_dim, K, 0.0f, ivb.data(), N);
        for (int r = 0; r < rows; r++)
          for (int n = 0; n < N; n++) y[static_cast<size_t>(r) * N + n] += ivb[n];
      }
      const std::vector<float> *z = NULL; const TimeSet *zt = NULL;
      if (d.bypass_layer != -2) { const int bi = d.bypass_layer < 0 ? n_layers : d.bypass_layer; z = &act[bi]; zt = &req[bi]; }
      for (int r = 0; r < rows; r++) {
        float *row = &y[static_cast<size_t>(r) * N];
        if (d.relu) for (int n = 0; n < N; n++) row[n] = row[n] < 0.0f ? 0.0f : row[n];
        if (d.bn_scale) for (int n = 0; n < N; n++) row[n] = row[n] * d.bn_scale[n] + d.bn_offset[n];
        if (z) {
          const float *zr = &(*z)[static_cast<size_t>(zt->Find(t[r])) * N];
          for (int n = 0; n < N; n++) row[n] += d.bypass_scale * zr[n];
        }
        if (d.log_softmax) {
          float mx = row[0], sum = 0.0f;
          for (int n = 1; n < N; n++) mx = std::max(mx, row[n]);
          for (int n = 0; n < N; n++) sum += expf((row[n] -= mx));
          sum = logf(sum);
          for (int n = 0; n < N; n++) row[n] -= sum;
        }
        if (d.post_offset) for (int n = 0; n < N; n++) row[n] += d.post_offset[n];
        if (d.post_scale != 1.0f) for (int n = 0; n < N; n++) row[n] *= d.post_scale;
      }
    }
    memcpy(out + static_cast<size_t>(start) * P, act[n_layers - 1].data(), sizeof(float) * static_cast<size_t>(num) * P);
  }
  return n_out;
}

extern "C" int orc_nnet_forward_blas(const kamd_layer_desc *L, int n_layers, int input_dim, int subsampling, const float *feats, int T,
                                     const float *ivector, int frames_per_chunk, void *sgemm_ptr, float *out, int out_rows_cap) {
  return ForwardBlas(L, n_layers, input_dim, subsampling, feats, T, ivector, NULL, 0, 0, 1, frames_per_chunk, sgemm_ptr, out, out_rows_cap);
}

extern "C" int orc_nnet_forward_blas_chunked(const kamd_layer_desc *L, int n_layers, int input_dim, int subsampling, const float *feats, int T,
                                             const float *online_ivectors, int n_iv_rows, int iv_dim, int ivector_period, int frames_per_chunk,
                                             void *sgemm_ptr, float *out, int out_rows_cap) {
  if (!online_ivectors || n_iv_rows <= 0 || ivector_period <= 0) return -1;
  int fpc = frames_per_chunk;
  if (fpc > 0 && fpc % subsampling != 0) fpc = subsampling * ((fpc + subsampling - 1) / subsampling);
  return ForwardBlas(L, n_layers, input_dim, subsampling, feats, T, NULL, online_ivectors, n_iv_rows, iv_dim, ivector_period, fpc, sgemm_ptr, out,
                     out_rows_cap);
}
