// OnlineSilenceWeighting (online2/online-ivector-feature.h:404-535, .cc:447-668) and the delta-weight queue of
// OnlineIvectorFeature (UpdateFrameWeights / UpdateStatsUntilFrameWeighted, .cc:159-174, 263-306): which frames
// the i-vector statistics should count less because the decoder's current best path calls them silence (or a
// transition-id runs for too long), re-decided every chunk as the traceback changes.  Host bookkeeping per
// stream: a few integers per decoded frame.  The traceback itself comes from the device
// (kamd_decoder_frame_tracebacks), the re-weighted statistics go back to it
// (kamd_ivector_stream_update_weighted_device).
#include <algorithm>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

#include "common.h"

namespace kamd {

struct SilenceWeighting {
  std::vector<unsigned char> tid_is_silence;     // TransitionIdToPhone(tid) is one of --silence-phones
  float silence_weight = 1.0f;
  int max_state_duration = -1;                   // the reference reads the BaseFloat option into an int32 (.cc:522)
  int subsampling = 1;
  // per decoded frame (decoder frame rate): the token and transition-id of the last traceback, the weight the
  // statistics currently hold for it
  std::vector<int32_t> token, tid;
  std::vector<float> current_weight;
  int frames_output_and_correct = 0;
  // OnlineIvectorFeature's side: delta weights not yet applied, lowest frame first
  typedef std::pair<int32_t, float> Delta;
  std::priority_queue<Delta, std::vector<Delta>, std::greater<Delta> > pending;
  int most_recent_frame_with_weight = -1;

  void Resize(size_t n) { token.resize(n, -1); tid.resize(n, -1); current_weight.resize(n, 0.0f); }
  void Reset() {
    token.clear(); tid.clear(); current_weight.clear(); frames_output_and_correct = 0;
    pending = decltype(pending)(); most_recent_frame_with_weight = -1;
  }

  // GetBeginFrame (.cc:521-571): normally frames_output_and_correct; earlier when the last untouched frame sits
  // in a run of one transition-id that the new frames make longer than max_state_duration
  int BeginFrame() const {
    const int max_dur = max_state_duration, foc = frames_output_and_correct;
    if (max_dur <= 0 || foc == 0) return foc;
    const int t_last = foc - 1, t_end = static_cast<int>(tid.size());
    const int run_tid = tid[t_last];
    const int lo = std::max(0, t_last - max_dur), hi = std::min(t_last + max_dur, t_end - 1);
    int t_lower = t_last, t_upper = t_last;
    while (t_lower > lo && tid[t_lower - 1] == run_tid) t_lower--;
    while (t_upper < hi && tid[t_upper + 1] == run_tid) t_upper++;
    if (t_upper - t_lower + 1 <= max_dur) return foc;
    if (t_last - t_lower + 1 > max_dur) return t_upper - max_dur;   // the old part alone was already over the limit
    return t_lower;
  }
};

}  // namespace kamd
using kamd::SilenceWeighting;

extern "C" {

kamd_silence_weighting *kamd_silence_weighting_create(const uint8_t *tid_is_silence, int n_tids, float silence_weight,
                                                      float max_state_duration, int frame_subsampling_factor) {
  if (!tid_is_silence || n_tids <= 0 || frame_subsampling_factor < 1) {
    kamd::SetError(KAMD_ERR_ARG, "silence weighting: needs the transition-id table and a frame-subsampling factor >= 1");
    return NULL;
  }
  SilenceWeighting *w = new SilenceWeighting();
  w->tid_is_silence.assign(tid_is_silence, tid_is_silence + n_tids);
  w->silence_weight = silence_weight; w->max_state_duration = static_cast<int>(max_state_duration);
  w->subsampling = frame_subsampling_factor;
  return reinterpret_cast<kamd_silence_weighting *>(w);
}

void kamd_silence_weighting_destroy(kamd_silence_weighting *h) { delete reinterpret_cast<SilenceWeighting *>(h); }

int kamd_silence_weighting_reset(kamd_silence_weighting *h) {
  reinterpret_cast<SilenceWeighting *>(h)->Reset();
  return KAMD_OK;
}

// ComputeCurrentTraceback (.cc:464-510).  tids[k] / tokens[k]: frame num_frames_decoded - 1 - k of the decoder's
// best path without final-probs (kamd_decoder_frame_tracebacks).  Walks back until a frame whose token is the one
// recorded last time: everything older is unchanged.
int kamd_silence_weighting_compute_traceback(kamd_silence_weighting *h, int num_frames_decoded, const int32_t *tids,
                                             const int32_t *tokens, int n) {
  SilenceWeighting *w = reinterpret_cast<SilenceWeighting *>(h);
  const int prev = static_cast<int>(w->tid.size());
  if (num_frames_decoded < 0 || n < 0 || n > num_frames_decoded) return kamd::SetError(KAMD_ERR_ARG, "silence weighting: bad traceback length");
  if (prev > num_frames_decoded && w->tid[num_frames_decoded] != -1) return kamd::SetError(KAMD_ERR_STATE, "Number of frames decoded decreased");
  // how far back the walk goes; checked before anything is recorded
  int depth = 0;
  for (int frame = num_frames_decoded - 1; frame >= 0; frame--, depth++) {
    if (depth >= n) return kamd::SetError(KAMD_ERR_ARG, "silence weighting: traceback ends at frame %d before reaching a known token", frame + 1);
    if (tids[depth] <= 0 || tids[depth] >= static_cast<int>(w->tid_is_silence.size()))
      return kamd::SetError(KAMD_ERR_ARG, "silence weighting: transition-id %d outside the model's 1..%d", tids[depth],
                            static_cast<int>(w->tid_is_silence.size()) - 1);
    if (frame < prev && w->token[frame] == tokens[depth]) break;
  }
  if (prev < num_frames_decoded) w->Resize(num_frames_decoded);
  for (int k = 0; k < depth; k++) {
    const int frame = num_frames_decoded - 1 - k;
    if (w->frames_output_and_correct > frame) w->frames_output_and_correct = frame;
    w->token[frame] = tokens[k]; w->tid[frame] = tids[k];
  }
  return KAMD_OK;
}

// GetDeltaWeights (.cc:573-668) followed by OnlineIvectorFeature::UpdateFrameWeights (.cc:159-174): the changes
// are queued; *n_deltas (may be NULL) says how many (input frame, delta) pairs this call produced.
int kamd_silence_weighting_get_delta_weights(kamd_silence_weighting *h, int num_frames_ready_in, int32_t *n_deltas) {
  SilenceWeighting *w = reinterpret_cast<SilenceWeighting *>(h);
  const int fs = w->subsampling, num_frames_ready = (num_frames_ready_in + fs - 1) / fs;
  if (n_deltas) *n_deltas = 0;
  if (static_cast<int>(w->tid.size()) < num_frames_ready) w->Resize(num_frames_ready);
  const int begin = w->BeginFrame(), frames_out = static_cast<int>(w->tid.size()) - begin;
  if (frames_out <= 0) return KAMD_OK;
  const float sw = w->silence_weight;
  const int max_dur = w->max_state_duration;
  std::vector<float> fw(frames_out, 1.0f);
  if (w->tid[begin] == -1) {
    // no traceback inside the window: repeat the last weight handed out, or call it all silence at the very start
    std::fill(fw.begin(), fw.end(), begin == 0 ? sw : w->current_weight[begin - 1]);
  } else {
    int run_start = 0;
    for (int o = 0; o < frames_out; o++) {
      const int t = w->tid[begin + o];
      if (t == -1) { fw[o] = fw[o - 1]; continue; }       // beyond the traceback: the newest decision carries on
      if (w->tid_is_silence[t]) fw[o] = sw;
      if (max_dur > 0 && (o + 1 == frames_out || t != w->tid[begin + o + 1])) {      // last frame of a run
        if (o - run_start + 1 >= max_dur) std::fill(fw.begin() + run_start, fw.begin() + o + 1, sw);
        if (o + 1 < frames_out) run_start = o + 1;
      }
    }
  }
  int produced = 0;
  for (int o = 0; o < frames_out; o++) {
    const int frame = begin + o;
    const float diff = fw[o] - w->current_weight[frame];
    w->current_weight[frame] = fw[o];
    if (diff != 0.0f || o + 1 == frames_out)              // the last frame is always reported, even unchanged
      for (int i = 0; i < fs; i++) {
        const int in_frame = frame * fs + i;
        w->pending.push(std::make_pair(in_frame, diff));
        w->most_recent_frame_with_weight = std::max(w->most_recent_frame_with_weight, in_frame);
        produced++;
      }
  }
  // NOT raised here: the header (.h:523-526) says GetDeltaWeights sets num_frames_output_and_correct_ to the number
  // of frames it output, but no line of the reference does -- it starts at 0 and is only ever lowered (.cc:501-502),
  // so GetBeginFrame always answers 0 and every call re-derives the weights of the whole utterance.  Same here.
  if (n_deltas) *n_deltas = produced;
  return KAMD_OK;
}

// The part of UpdateStatsUntilFrameWeighted (.cc:263-306) that decides WHAT enters the statistics when the estimate
// is advanced to input frame `frame`: every queued delta for frames <= frame, duplicates of a frame summed and
// zero sums dropped (MergePairVectorSumming, util/stl-utils.h:290-315), in increasing frame order.
int kamd_silence_weighting_pop_until(kamd_silence_weighting *h, int frame, int32_t *frames, float *weights, int cap, int32_t *n) {
  SilenceWeighting *w = reinterpret_cast<SilenceWeighting *>(h);
  *n = 0;
  if (frame > w->most_recent_frame_with_weight)
    return kamd::SetError(KAMD_ERR_STATE, "silence weighting: no weight was provided for frame %d (most recent: %d)", frame,
                          w->most_recent_frame_with_weight);
  bool open = false;
  int32_t cur = -1; float sum = 0.0f;
  auto flush = [&]() -> int {
    if (open && sum != 0.0f) {
      if (*n >= cap) return kamd::SetError(KAMD_ERR_ARG, "silence weighting: room for %d weighted frames is not enough", cap);
      frames[*n] = cur; weights[*n] = sum; (*n)++;
    }
    return KAMD_OK;
  };
  while (!w->pending.empty() && w->pending.top().first <= frame) {
    const SilenceWeighting::Delta d = w->pending.top();
    w->pending.pop();
    if (open && d.first == cur) { sum += d.second; continue; }
    if (flush() != KAMD_OK) return KAMD_ERR_ARG;
    open = true; cur = d.first; sum = d.second;
  }
  return flush();
}

int kamd_silence_weighting_num_pending(const kamd_silence_weighting *h) {
  return static_cast<int>(reinterpret_cast<const SilenceWeighting *>(h)->pending.size());
}

}  // extern "C"
