// oracle/orc_nnet.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
//
// CPU restatement of the nnet3 inference slice for TDNN(-F) chain models:
// what DecodableNnetSimple::GetOutputForFrame (nnet3/nnet-am-decodable-simple.cc:
// 93-276) returns for every subsampled frame, evaluated the most naive way: a
// memoised recursion over (layer, time) with scalar dot products.
// PARITY UNPINNED against reference-run outputs: nnet3 cannot be built in this image
// (base/kaldi-types.h:44 needs <fst/types.h>, base/version.h is generated, no
// CBLAS headers), and nnet3's own tests are randomized self-consistency tests with
// no golden vectors (nnet3/nnet-compute-test.cc).  It is cross-checked against an
// independent float64 numpy evaluation in tests/test_oracle_nnet.py.
//
// Component semantics restated:
//   TdnnComponent::Propagate      nnet3/nnet-tdnn-component.cc:181-212
//   AffineComponent::Propagate    nnet3/nnet-simple-component.cc:1234-1243
//   RectifiedLinearComponent      nnet3/nnet-simple-component.cc:957-965
//   BatchNormComponent test mode  nnet3/nnet-normalize-component.cc:453-464
//   tdnnf bypass Sum(Scale(s,x),y) steps/libs/nnet3/xconfig/composite_layers.py:201-215
//   -log_priors, *acoustic_scale  nnet3/nnet-am-decodable-simple.cc:268-271
//   edge clamping of input frames nnet3/nnet-am-decodable-simple.cc:147-160
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../include/kaldi_amd.h"

namespace {

struct Eval {
  const kamd_layer_desc *L;
  int n_layers, input_dim, T;
  const float *feats, *ivector;
  // looped-decodable i-vectors: row t of the first layer reads table row clamp(floor(t / period) - slot_first)
  const float *slot_table = NULL; int slot_period = 0, slot_first = 0, slot_count = 0, iv_dim = 0;
  const float *IvectorAt(int t) const {
    if (slot_period <= 0) return ivector;
    int sl = (t >= 0 ? t / slot_period : -((-t + slot_period - 1) / slot_period)) - slot_first;
    sl = sl < 0 ? 0 : (sl >= slot_count ? slot_count - 1 : sl);
    return slot_table + static_cast<int64_t>(sl) * iv_dim;
  }
  std::vector<std::unordered_map<int, std::vector<float> > > memo;

  const float *Input(int t) const {
    if (t < 0) t = 0;
    if (t >= T) t = T - 1;
    return feats + static_cast<int64_t>(t) * input_dim;
  }
  const float *Get(int layer, int t) {
    if (layer == -1) return Input(t);
    std::unordered_map<int, std::vector<float> > &m = memo[layer];
    std::unordered_map<int, std::vector<float> >::iterator it = m.find(t);
    if (it != m.end()) return it->second.data();
    const kamd_layer_desc &l = L[layer];
    // slices of the input: all from input_layer, in_dim wide -- or (multi_input: Append over different producers,
    // nnet-compute.cc:309-383 kCopyRows over arbitrary sources) each from its own layer with its own width
    std::vector<int> sdim(l.n_offsets), scol(l.n_offsets);
    int K = l.ivector_dim;
    for (int i = 0; i < l.n_offsets; i++) { sdim[i] = l.multi_input ? l.slice_dim[i] : l.in_dim; scol[i] = K - l.ivector_dim; K += sdim[i]; }
    std::vector<float> y(l.out_dim);
    std::vector<const float *> xs(l.n_offsets);
    for (int i = 0; i < l.n_offsets; i++) xs[i] = Get(l.multi_input ? l.slice_layer[i] : l.input_layer, t + l.offsets[i]);
    const float *z = NULL;
    if (l.bypass_layer != -2) z = Get(l.bypass_layer, t);
    for (int o = 0; o < l.out_dim; o++) {
      const float *w = l.W + static_cast<int64_t>(o) * K;
      float acc = l.bias ? l.bias[o] : 0.0f;   // bias first, then AddMatMat (:189-210)
      for (int i = 0; i < l.n_offsets; i++) {
        const float *x = xs[i], *wi = w + scol[i];
        float s = 0.0f;
        for (int k = 0; k < sdim[i]; k++) s += wi[k] * x[k];
        acc += s;
      }
      if (l.ivector_dim > 0) {
        const float *wi = w + (K - l.ivector_dim);
        float s = 0.0f;
        const float *iv = IvectorAt(t);
        for (int k = 0; k < l.ivector_dim; k++) s += wi[k] * iv[k];
        acc += s;
      }
      if (l.relu && acc < 0.0f) acc = 0.0f;
      if (l.bn_scale) acc = acc * l.bn_scale[o] + l.bn_offset[o];
      if (z) acc += l.bypass_scale * z[o];
      y[o] = acc;
    }
    if (l.log_softmax) {   // VectorBase::ApplyLogSoftMax (matrix/kaldi-vector.cc:876-884), per row
      float mx = y[0], sum = 0.0f;
      for (int o = 1; o < l.out_dim; o++) mx = std::max(mx, y[o]);
      for (int o = 0; o < l.out_dim; o++) sum += expf((y[o] -= mx));
      sum = logf(sum);
      for (int o = 0; o < l.out_dim; o++) y[o] -= sum;
    }
    for (int o = 0; o < l.out_dim; o++) {
      float acc = y[o];
      if (l.post_offset) acc += l.post_offset[o];
      acc *= l.post_scale;
      y[o] = acc;
    }
    std::vector<float> &slot = memo[layer][t];
    slot.swap(y);
    return slot.data();
  }
};

void Context(const kamd_layer_desc *L, int layer, int *left, int *right) {
  // ComputeSimpleNnetContext (nnet3/nnet-utils.cc:146) for a layer chain.
  if (layer == -1) { *left = 0; *right = 0; return; }
  const kamd_layer_desc &l = L[layer];
  *left = 0; *right = 0;
  int il = 0, ir = 0;
  if (!l.multi_input) Context(L, l.input_layer, &il, &ir);       // (once per layer: the recursion is a chain)
  for (int i = 0; i < l.n_offsets; i++) {
    if (l.multi_input) Context(L, l.slice_layer[i], &il, &ir);
    *left = std::max(*left, il - std::min(0, l.offsets[i]));
    *right = std::max(*right, ir + std::max(0, l.offsets[i]));
  }
  if (l.bypass_layer != -2) {
    int bl, br;
    Context(L, l.bypass_layer, &bl, &br);
    *left = std::max(*left, bl);
    *right = std::max(*right, br);
  }
}

}  // namespace

extern "C" {

void orc_nnet_context(const kamd_layer_desc *layers, int n_layers, int *left, int *right) {
  Context(layers, n_layers - 1, left, right);
}

int orc_nnet_forward(const kamd_layer_desc *layers, int n_layers, int input_dim,
                     int subsampling, const float *feats, int T, const float *ivector,
                     float *out, int out_rows_cap) {
  if (T <= 0) return 0;
  int n_out = (T + subsampling - 1) / subsampling;  // nnet-am-decodable-simple.cc:44-46
  if (n_out > out_rows_cap) return -1;
  Eval e;
  e.L = layers; e.n_layers = n_layers; e.input_dim = input_dim; e.T = T;
  e.feats = feats; e.ivector = ivector;
  e.memo.resize(n_layers);
  int P = layers[n_layers - 1].out_dim;
  for (int i = 0; i < n_out; i++) {
    const float *y = e.Get(n_layers - 1, i * subsampling);
    memcpy(out + static_cast<int64_t>(i) * P, y, sizeof(float) * P);
  }
  return n_out;
}


// DecodableNnetLoopedOnline's i-vector semantics over a whole utterance: Round(ivector, period)
// (nnet3/nnet-compile-looped.cc:186-207, ModifyNnetIvectorPeriod): the first layer's row at time t reads
// slot floor(t / period); slot_table holds slots slot_first .. slot_first + slot_count - 1.
int orc_nnet_forward_slots(const kamd_layer_desc *layers, int n_layers, int input_dim, int subsampling, const float *feats, int T,
                           const float *slot_table, int slot_first, int slot_count, int iv_dim, int period, float *out,
                           int out_rows_cap) {
  if (T <= 0) return 0;
  int n_out = (T + subsampling - 1) / subsampling;
  if (n_out > out_rows_cap) return -1;
  Eval e;
  e.L = layers; e.n_layers = n_layers; e.input_dim = input_dim; e.T = T;
  e.feats = feats; e.ivector = NULL;
  e.slot_table = slot_table; e.slot_period = period; e.slot_first = slot_first; e.slot_count = slot_count; e.iv_dim = iv_dim;
  e.memo.resize(n_layers);
  int P = layers[n_layers - 1].out_dim;
  for (int i = 0; i < n_out; i++) memcpy(out + static_cast<int64_t>(i) * P, e.Get(n_layers - 1, i * subsampling), sizeof(float) * P);
  return n_out;
}

// DecodableNnetSimple with online ivectors (nnet3/nnet-am-decodable-simple.cc:93-214): the
// decoder asks for frames in order, so chunks start at 0, C, 2C, ... with C =
// frames_per_chunk / subsampling after CheckAndFixConfigs (:278-310) rounded frames_per_chunk
// up to a multiple of the subsampling factor; every chunk is evaluated with ONE ivector, the
// row GetCurrentIvector (:181-211) picks for the middle of the chunk.
int orc_nnet_forward_chunked(const kamd_layer_desc *layers, int n_layers, int input_dim, int subsampling,
                             const float *feats, int T, const float *online_ivectors, int n_iv_rows, int iv_dim,
                             int ivector_period, int frames_per_chunk, float *out, int out_rows_cap) {
  if (T <= 0) return 0;
  const int n_out = (T + subsampling - 1) / subsampling;
  if (n_out > out_rows_cap) return -1;
  if (frames_per_chunk % subsampling != 0) frames_per_chunk = subsampling * ((frames_per_chunk + subsampling - 1) / subsampling);
  const int C = frames_per_chunk / subsampling;
  const int P = layers[n_layers - 1].out_dim;
  for (int start = 0; start < n_out; start += C) {
    const int num = std::min(n_out - start, C);
    const int first_output_frame = start * subsampling, last_output_frame = (start + num - 1) * subsampling;
    const int frame_to_search = first_output_frame + (last_output_frame - first_output_frame) / 2;
    int ivector_frame = frame_to_search / ivector_period;
    if (ivector_frame >= n_iv_rows) {
      if ((ivector_frame - (n_iv_rows - 1)) * ivector_period > 50) return -2;   // "Could not get iVector for frame"
      ivector_frame = n_iv_rows - 1;
    }
    Eval e;
    e.L = layers; e.n_layers = n_layers; e.input_dim = input_dim; e.T = T;
    e.feats = feats; e.ivector = online_ivectors + static_cast<int64_t>(ivector_frame) * iv_dim;
    e.memo.resize(n_layers);
    for (int i = 0; i < num; i++) {
      const float *y = e.Get(n_layers - 1, (start + i) * subsampling);
      memcpy(out + static_cast<int64_t>(start + i) * P, y, sizeof(float) * P);
    }
  }
  return n_out;
}

// NnetBatchComputer::SplitUtteranceIntoTasks + ComputeSimple + MergeTaskOutput for ONE utterance with online i-vectors
// (nnet3/nnet-batch-compute.cc): what nnet3-latgen-faster-batch evaluates, statement by statement --
//   GetOutputFrameInfoForTasks :586-668  chunks of fpc = frames_per_chunk / f subsampled frames (integer division, no
//       rounding up); the LAST chunk ends on the utterance's last frame and overlaps the one before it
//       (num_initial_unused_output_frames); an utterance shorter than a chunk is ONE chunk of fpc frames
//       (ensure_exact_final_context = false, the default);
//   SplitInputToTasks :705-770  every task's input is [begin_output_t * f - left, end_output_t * f + right), clamped to
//       the utterance (extra contexts 0, the defaults);
//   AddOnlineIvectorsToTasks :670-703  the row of the task's middle: (begin_output_t + num_output_frames / 2) * f /
//       period, the last row when that is at most 20 input frames beyond the table, an error otherwise;
//   MergeTaskOutput :832-870  rows [num_initial_unused, + num_used) of every task's output, in order.
// Every task is evaluated WHOLE here (all fpc output frames, the unused and the padded ones included), as the reference
// does; the device evaluates the used rows only and must give the same numbers.  tasks_out (optional, 6 ints per task):
// first_used_output_frame_index, num_initial_unused_output_frames, num_used_output_frames, num_output_frames,
// first_input_t, ivector row.  Returns the number of output rows, -1: buffer too small, -2: no i-vector for a task.
int orc_nnet_forward_batch_computer(const kamd_layer_desc *layers, int n_layers, int input_dim, int subsampling,
                                    int left_context, int right_context,
                                    const float *feats, int T, const float *online_ivectors, int n_iv_rows, int iv_dim,
                                    int ivector_period, int frames_per_chunk, float *out, int out_rows_cap,
                                    int *tasks_out, int tasks_cap, int *n_tasks_out) {
  if (T <= 0) return 0;
  const int f = subsampling;
  const int num_subsampled_frames = (T + f - 1) / f;
  if (num_subsampled_frames > out_rows_cap) return -1;
  const int fpc = frames_per_chunk / f;
  if (fpc <= 0) return -3;
  const int num_tasks = (num_subsampled_frames + fpc - 1) / fpc;
  struct Task { int first_used, unused, used, num_out, first_input_t, iv_row; };
  std::vector<Task> tasks(num_tasks);
  if (num_subsampled_frames <= fpc) {
    tasks[0].first_used = 0; tasks[0].num_out = fpc; tasks[0].unused = 0; tasks[0].used = num_subsampled_frames;
  } else {
    for (int i = 0; i + 1 < num_tasks; i++) { tasks[i].num_out = fpc; tasks[i].unused = 0; tasks[i].used = fpc; tasks[i].first_used = i * fpc; }
    Task &t = tasks[num_tasks - 1];
    t.num_out = fpc;
    t.unused = (num_tasks - 1) * fpc - (num_subsampled_frames - fpc);
    t.used = num_subsampled_frames - (num_tasks - 1) * fpc;
    t.first_used = (num_tasks - 1) * fpc;
  }
  const int P = layers[n_layers - 1].out_dim;
  int cur = 0;
  for (int i = 0; i < num_tasks; i++) {
    Task &t = tasks[i];
    const int begin_output_t = t.first_used - t.unused, end_output_t = begin_output_t + t.num_out;
    const int begin_input_t = begin_output_t * f, end_input_t = end_output_t * f;
    const int begin_padded = begin_input_t - left_context, end_padded = end_input_t + right_context;
    t.first_input_t = begin_padded - begin_output_t * f;
    // the task's own copy of its input rows, clamped like SplitInputToTasks (:755-764)
    const int rows = end_padded - begin_padded;
    std::vector<float> in(static_cast<size_t>(rows) * input_dim);
    for (int tt = begin_padded; tt < end_padded; tt++) {
      const int tc = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
      memcpy(&in[static_cast<size_t>(tt - begin_padded) * input_dim], feats + static_cast<int64_t>(tc) * input_dim, sizeof(float) * input_dim);
    }
    if (online_ivectors) {
      const int mid_output_t = begin_output_t + t.num_out / 2, mid_input_t = mid_output_t * f;
      int ivector_frame = mid_input_t / ivector_period;
      const int margin_in_ivector_frames = (20 + ivector_period - 1) / ivector_period;
      if (ivector_frame >= n_iv_rows) {
        if (n_iv_rows > 0 && ivector_frame > n_iv_rows - margin_in_ivector_frames) ivector_frame = n_iv_rows - 1;
        else return -2;
      }
      t.iv_row = ivector_frame;
    } else t.iv_row = -1;
    // the task's computation sees ONLY its own rows: times are relative to its first row (the clamp of Eval::Input then
    // never binds differently from the copy above, which already holds the clamped rows)
    Eval e;
    e.L = layers; e.n_layers = n_layers; e.input_dim = input_dim; e.T = rows;
    e.feats = in.data(); e.ivector = online_ivectors ? online_ivectors + static_cast<int64_t>(t.iv_row) * iv_dim : NULL;
    e.memo.resize(n_layers);
    for (int r = 0; r < t.used; r++) {
      const int o = t.unused + r;                                    // row of the task's output
      const float *y = e.Get(n_layers - 1, left_context + o * f);    // output frame o of the task sits at input row left + o f
      if (cur != t.first_used + r) return -4;
      memcpy(out + static_cast<int64_t>(cur) * P, y, sizeof(float) * P);
      cur++;
    }
    if (tasks_out && i < tasks_cap) {
      int *w = tasks_out + 6 * i;
      w[0] = t.first_used; w[1] = t.unused; w[2] = t.used; w[3] = t.num_out; w[4] = t.first_input_t; w[5] = t.iv_row;
    }
  }
  if (n_tasks_out) *n_tasks_out = num_tasks;
  return cur == num_subsampled_frames ? cur : -4;
}

}  // extern "C"
