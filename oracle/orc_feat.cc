// oracle/orc_feat.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle; never shipped,
// never measured as the product).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load it.
//
// CPU restatement of Kaldi's MFCC / fbank computation, plain scalar C++.
// Pinned against the reference's HTK golden vectors (feat/test_data/*, copied as
// data to tests/golden/feat/) by tests/test_oracle_feat.py.
//
// Every function cites the reference file:line (relative to src/) it follows.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/kaldi_amd.h"

namespace {

// feat/feature-window.h:118-130 (WindowShift/WindowSize/PaddedWindowSize).
int WindowShift(const kamd_frame_opts &o) {
  return static_cast<int>(o.samp_freq * 0.001f * o.frame_shift_ms);
}
int WindowSize(const kamd_frame_opts &o) {
  return static_cast<int>(o.samp_freq * 0.001f * o.frame_length_ms);
}
int RoundUpPow2(int n) {  // base/kaldi-math.cc RoundUpToNearestPowerOfTwo
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}
int PaddedWindowSize(const kamd_frame_opts &o) {
  return o.round_to_power_of_two ? RoundUpPow2(WindowSize(o)) : WindowSize(o);
}

// feat/feature-window.cc:28-39.
int64_t FirstSampleOfFrame(int frame, const kamd_frame_opts &o) {
  int64_t shift = WindowShift(o);
  if (o.snip_edges) return frame * shift;
  int64_t mid = shift * frame + shift / 2;
  return mid - WindowSize(o) / 2;
}

// feat/feature-window.cc:41-87 (flush == true: offline).
int NumFrames(int64_t num_samples, const kamd_frame_opts &o) {
  int64_t shift = WindowShift(o), len = WindowSize(o);
  if (o.snip_edges) {
    if (num_samples < len) return 0;
    return static_cast<int>(1 + (num_samples - len) / shift);
  }
  return static_cast<int>((num_samples + shift / 2) / shift);
}

// feat/feature-window.cc:109-131.
std::vector<float> WindowFunction(const kamd_frame_opts &o) {
  int n = WindowSize(o);
  std::vector<float> w(n);
  double a = 2.0 * M_PI / (n - 1);
  for (int i = 0; i < n; i++) {
    double x = i;
    switch (o.window_type) {
      case KAMD_WIN_HANNING: w[i] = 0.5 - 0.5 * cos(a * x); break;
      case KAMD_WIN_HAMMING: w[i] = 0.54 - 0.46 * cos(a * x); break;
      case KAMD_WIN_POVEY: w[i] = pow(0.5 - 0.5 * cos(a * x), 0.85); break;
      case KAMD_WIN_RECTANGULAR: w[i] = 1.0; break;
      default:
        w[i] = o.blackman_coeff - 0.5 * cos(a * x) +
               (0.5 - o.blackman_coeff) * cos(2 * a * x);
    }
  }
  return w;
}

// feat/feature-window.cc:162-220 ExtractWindow + :133-156 ProcessWindow
// (dither must be 0: the reference's dither is random, feature-window.cc:90-98).
void ExtractWindow(const float *wave, int64_t n, int f, const kamd_frame_opts &o,
                   const std::vector<float> &win, std::vector<float> *window,
                   float *log_energy_pre_window) {
  int len = WindowSize(o), padded = PaddedWindowSize(o);
  int64_t start = FirstSampleOfFrame(f, o);
  window->assign(padded, 0.0f);
  for (int s = 0; s < len; s++) {
    int64_t i = s + start;
    while (i < 0 || i >= n) {  // reflection, :198-206
      if (i < 0) i = -i - 1;
      else i = 2 * n - 1 - i;
    }
    (*window)[s] = wave[i];
  }
  float *x = window->data();
  if (o.remove_dc_offset) {  // :143-144
    float sum = 0;
    for (int i = 0; i < len; i++) sum += x[i];
    float m = -sum / len;
    for (int i = 0; i < len; i++) x[i] += m;
  }
  if (log_energy_pre_window) {  // :146-150
    float e = 0;
    for (int i = 0; i < len; i++) e += x[i] * x[i];
    e = std::max(e, std::numeric_limits<float>::epsilon());
    *log_energy_pre_window = logf(e);
  }
  if (o.preemph_coeff != 0.0f) {  // :100-107 Preemphasize
    for (int i = len - 1; i > 0; i--) x[i] -= o.preemph_coeff * x[i - 1];
    x[0] -= o.preemph_coeff * x[0];
  }
  for (int i = 0; i < len; i++) x[i] *= win[i];  // :155
}

// Complex radix-2 FFT (float), standing in for SplitRadixRealFft::Compute
// (matrix/srfft.cc:356); only |X_k|^2 for k = 0..N/2 is consumed
// (feat/feature-functions.cc:29-51 ComputePowerSpectrum), so the packing
// convention of srfft is irrelevant here.
void PowerSpectrum(const std::vector<float> &x, std::vector<float> *power) {
  int N = static_cast<int>(x.size());
  std::vector<float> re(N), im(N, 0.0f);
  int bits = 0;
  while ((1 << bits) < N) bits++;
  for (int i = 0; i < N; i++) {
    int r = 0;
    for (int b = 0; b < bits; b++)
      if (i & (1 << b)) r |= 1 << (bits - 1 - b);
    re[r] = x[i];
  }
  for (int len = 2; len <= N; len <<= 1) {
    int half = len / 2;
    for (int k = 0; k < half; k++) {
      double ang = -2.0 * M_PI * k / len;
      float wr = static_cast<float>(cos(ang)), wi = static_cast<float>(sin(ang));
      for (int s = k; s < N; s += len) {
        int t = s + half;
        float tr = re[t] * wr - im[t] * wi, ti = re[t] * wi + im[t] * wr;
        re[t] = re[s] - tr; im[t] = im[s] - ti;
        re[s] += tr; im[s] += ti;
      }
    }
  }
  power->resize(N / 2 + 1);
  for (int k = 0; k <= N / 2; k++) (*power)[k] = re[k] * re[k] + im[k] * im[k];
}

// feat/mel-computations.h:81-87.
inline float MelScale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
inline float InverseMelScale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

// feat/mel-computations.cc:142-200 VtlnWarpFreq / VtlnWarpMelFreq.
float VtlnWarpFreq(float vtln_low_cutoff, float vtln_high_cutoff, float low_freq,
                   float high_freq, float vtln_warp_factor, float freq) {
  if (freq < low_freq || freq > high_freq) return freq;
  float one = 1.0f;
  float l = vtln_low_cutoff * std::max(one, vtln_warp_factor);
  float h = vtln_high_cutoff * std::min(one, vtln_warp_factor);
  float scale = 1.0f / vtln_warp_factor;
  float Fl = scale * l, Fh = scale * h;
  float scale_left = (Fl - low_freq) / (l - low_freq);
  float scale_right = (high_freq - Fh) / (high_freq - h);
  if (freq < l) return low_freq + scale_left * (freq - low_freq);
  else if (freq < h) return scale * freq;
  else return high_freq + scale_right * (freq - high_freq);
}
float VtlnWarpMelFreq(float vl, float vh, float lf, float hf, float warp, float mel) {
  return MelScale(VtlnWarpFreq(vl, vh, lf, hf, warp, InverseMelScale(mel)));
}

struct MelBanks {  // feat/mel-computations.cc:33-133
  std::vector<int> first;
  std::vector<std::vector<float> > w;
  bool htk_mode;
  MelBanks(const kamd_mel_opts &o, const kamd_frame_opts &fo, float warp)
      : htk_mode(o.htk_mode != 0) {
    int num_bins = o.num_bins;
    float sample_freq = fo.samp_freq;
    int padded = PaddedWindowSize(fo);
    int num_fft_bins = padded / 2;
    float nyquist = 0.5f * sample_freq;
    float low_freq = o.low_freq, high_freq;
    if (o.high_freq > 0.0f) high_freq = o.high_freq;
    else high_freq = nyquist + o.high_freq;
    float fft_bin_width = sample_freq / padded;
    float mel_low = MelScale(low_freq), mel_high = MelScale(high_freq);
    float mel_delta = (mel_high - mel_low) / (num_bins + 1);
    float vtln_low = o.vtln_low, vtln_high = o.vtln_high;
    if (vtln_high < 0.0f) vtln_high += nyquist;
    first.resize(num_bins);
    w.resize(num_bins);
    for (int bin = 0; bin < num_bins; bin++) {
      float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta,
            right = mel_low + (bin + 2) * mel_delta;
      if (warp != 1.0f) {
        left = VtlnWarpMelFreq(vtln_low, vtln_high, low_freq, high_freq, warp, left);
        center = VtlnWarpMelFreq(vtln_low, vtln_high, low_freq, high_freq, warp, center);
        right = VtlnWarpMelFreq(vtln_low, vtln_high, low_freq, high_freq, warp, right);
      }
      std::vector<float> this_bin(num_fft_bins, 0.0f);
      int fi = -1, li = -1;
      for (int i = 0; i < num_fft_bins; i++) {
        float freq = fft_bin_width * i;
        float mel = MelScale(freq);
        if (mel > left && mel < right) {
          float weight;
          if (mel <= center) weight = (mel - left) / (center - left);
          else weight = (right - mel) / (right - center);
          this_bin[i] = weight;
          if (fi == -1) fi = i;
          li = i;
        }
      }
      first[bin] = fi;
      w[bin].assign(this_bin.begin() + fi, this_bin.begin() + li + 1);
      if (o.htk_mode && bin == 0 && mel_low != 0.0f) w[bin][0] = 0.0f;  // :121-123
    }
  }
  // feat/mel-computations.cc:226-252.
  void Compute(const std::vector<float> &power, float *out) const {
    for (size_t i = 0; i < w.size(); i++) {
      float e = 0;
      for (size_t j = 0; j < w[i].size(); j++) e += w[i][j] * power[first[i] + j];
      if (htk_mode && e < 1.0f) e = 1.0f;
      out[i] = e;
    }
  }
};

}  // namespace

extern "C" {

int orc_feat_num_frames(const kamd_frame_opts *o, int64_t n) { return NumFrames(n, *o); }

// feat/feature-mfcc.cc:28-80 MfccComputer::Compute over
// feat/feature-common-inl.h:60-83 OfflineFeatureTpl::Compute.
int orc_mfcc_compute(const kamd_mfcc_opts *op, float vtln_warp, const float *wave,
                     int64_t n, float *out, int out_rows_cap) {
  const kamd_frame_opts &fo = op->frame;
  if (fo.dither != 0.0f) return -1;
  int T = NumFrames(n, fo);
  if (T > out_rows_cap) return -1;
  int num_bins = op->mel.num_bins, C = op->num_ceps;
  std::vector<float> win = WindowFunction(fo);
  MelBanks banks(op->mel, fo, vtln_warp);
  // matrix/matrix-functions.cc:592-608 ComputeDctMatrix (first num_ceps rows).
  std::vector<float> dct(C * num_bins);
  {
    float norm0 = std::sqrt(1.0 / static_cast<float>(num_bins));
    for (int j = 0; j < num_bins; j++) dct[j] = norm0;
    float norm = std::sqrt(2.0 / static_cast<float>(num_bins));
    for (int k = 1; k < C; k++)
      for (int m = 0; m < num_bins; m++)
        dct[k * num_bins + m] =
            norm * std::cos(static_cast<double>(M_PI) / num_bins * (m + 0.5) * k);
  }
  std::vector<float> lifter(C, 1.0f);  // feat/mel-computations.cc:253-259
  if (op->cepstral_lifter != 0.0f)
    for (int i = 0; i < C; i++)
      lifter[i] = 1.0 + 0.5 * op->cepstral_lifter * sin(M_PI * i / op->cepstral_lifter);
  float log_energy_floor = op->energy_floor > 0.0f ? logf(op->energy_floor) : 0.0f;
  bool need_raw = op->use_energy && op->raw_energy;
  std::vector<float> window, power, mel(num_bins);
  for (int f = 0; f < T; f++) {
    float raw_log_energy = 0.0f;
    ExtractWindow(wave, n, f, fo, win, &window, need_raw ? &raw_log_energy : NULL);
    float signal_log_energy = raw_log_energy;
    if (op->use_energy && !op->raw_energy) {  // feature-mfcc.cc:37-39
      float e = 0;
      for (size_t i = 0; i < window.size(); i++) e += window[i] * window[i];
      signal_log_energy = logf(std::max(e, std::numeric_limits<float>::min()));
    }
    PowerSpectrum(window, &power);
    banks.Compute(power, mel.data());
    for (int i = 0; i < num_bins; i++)
      mel[i] = logf(std::max(mel[i], std::numeric_limits<float>::epsilon()));
    float *feat = out + static_cast<int64_t>(f) * C;
    for (int k = 0; k < C; k++) {
      float s = 0;
      for (int m = 0; m < num_bins; m++) s += dct[k * num_bins + m] * mel[m];
      feat[k] = s;
    }
    if (op->cepstral_lifter != 0.0f)
      for (int k = 0; k < C; k++) feat[k] *= lifter[k];
    if (op->use_energy) {
      if (op->energy_floor > 0.0f && signal_log_energy < log_energy_floor)
        signal_log_energy = log_energy_floor;
      feat[0] = signal_log_energy;
    }
    if (op->htk_compat) {  // feature-mfcc.cc:69-79
      float energy = feat[0];
      for (int i = 0; i < C - 1; i++) feat[i] = feat[i + 1];
      if (!op->use_energy) energy *= M_SQRT2;
      feat[C - 1] = energy;
    }
  }
  return T;
}

// feat/feature-fbank.cc:73-122 FbankComputer::Compute.
int orc_fbank_compute(const kamd_fbank_opts *op, float vtln_warp, const float *wave,
                      int64_t n, float *out, int out_rows_cap) {
  const kamd_frame_opts &fo = op->frame;
  if (fo.dither != 0.0f) return -1;
  int T = NumFrames(n, fo);
  if (T > out_rows_cap) return -1;
  int num_bins = op->mel.num_bins;
  int dim = num_bins + (op->use_energy ? 1 : 0);
  std::vector<float> win = WindowFunction(fo);
  MelBanks banks(op->mel, fo, vtln_warp);
  float log_energy_floor = op->energy_floor > 0.0f ? logf(op->energy_floor) : 0.0f;
  bool need_raw = op->use_energy && op->raw_energy;
  std::vector<float> window, power;
  for (int f = 0; f < T; f++) {
    float raw_log_energy = 0.0f;
    ExtractWindow(wave, n, f, fo, win, &window, need_raw ? &raw_log_energy : NULL);
    float signal_log_energy = raw_log_energy;
    if (op->use_energy && !op->raw_energy) {
      float e = 0;
      for (size_t i = 0; i < window.size(); i++) e += window[i] * window[i];
      signal_log_energy = logf(std::max(e, std::numeric_limits<float>::min()));
    }
    PowerSpectrum(window, &power);
    if (!op->use_power)
      for (size_t i = 0; i < power.size(); i++) power[i] = powf(power[i], 0.5f);
    float *feat = out + static_cast<int64_t>(f) * dim;
    int mel_offset = (op->use_energy && !op->htk_compat) ? 1 : 0;
    banks.Compute(power, feat + mel_offset);
    if (op->use_log_fbank)
      for (int i = 0; i < num_bins; i++)
        feat[mel_offset + i] =
            logf(std::max(feat[mel_offset + i], std::numeric_limits<float>::epsilon()));
    if (op->use_energy) {
      if (op->energy_floor > 0.0f && signal_log_energy < log_energy_floor)
        signal_log_energy = log_energy_floor;
      feat[op->htk_compat ? num_bins : 0] = signal_log_energy;
    }
  }
  return T;
}

}  // extern "C"

// ---- CMVN (test infrastructure, like the rest of this file): AccCmvnStats (transform/cmvn.cc:30-62) and
// ApplyCmvn (transform/cmvn.cc:64-118) restated.  PARITY UNPINNED: the reference holds no test for them.
extern "C" void orc_cmvn_acc_stats(const float *feats, int T, int dim, double *stats /* [2][dim+1], added to */) {
  for (int t = 0; t < T; t++) {
    stats[dim] += 1.0f;
    for (int k = 0; k < dim; k++) {
      const float x = feats[static_cast<size_t>(t) * dim + k];
      stats[k] += x * 1.0f;
      stats[dim + 1 + k] += x * x * 1.0f;
    }
  }
}
// AccCmvnStats(feats, &weights, stats) (transform/cmvn.cc:30-62)
extern "C" void orc_cmvn_acc_stats_weighted(const float *feats, int T, int dim, const float *weights, double *stats) {
  for (int t = 0; t < T; t++) {
    const float weight = weights[t];
    if (weight == 0.0f) continue;
    stats[dim] += weight;
    for (int k = 0; k < dim; k++) {
      const float x = feats[static_cast<size_t>(t) * dim + k];
      stats[k] += x * weight;
      stats[dim + 1 + k] += x * x * weight;
    }
  }
}
extern "C" int orc_cmvn_apply(const double *stats, int var_norm, float *feats, int T, int dim) {
  const double count = stats[dim];
  if (count < 1.0) return -1;
  for (int k = 0; k < dim; k++) {
    float offset, scale = 1.0f;
    if (!var_norm) offset = static_cast<float>(-1.0 / count * stats[k]);
    else {
      const double mean = stats[k] / count;
      double var = stats[dim + 1 + k] / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      const double s = 1.0 / sqrt(var);
      offset = static_cast<float>(-(mean * s)); scale = static_cast<float>(s);
    }
    for (int t = 0; t < T; t++) {
      float x = feats[static_cast<size_t>(t) * dim + k];
      if (var_norm) x = x * scale;
      feats[static_cast<size_t>(t) * dim + k] = x + offset;
    }
  }
  return 0;
}

// ApplyCmvnReverse (transform/cmvn.cc:120-168)
extern "C" int orc_cmvn_apply_reverse(const double *stats, int var_norm, float *feats, int T, int dim) {
  const double count = stats[dim];
  if (count < 1.0) return -1;
  for (int k = 0; k < dim; k++) {
    const double mean = stats[k] / count;
    double scale = 1.0;
    if (var_norm) {
      double var = stats[dim + 1 + k] / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      scale = sqrt(var);
    }
    const float offset_f = static_cast<float>(mean), scale_f = static_cast<float>(scale);
    for (int t = 0; t < T; t++) {
      float x = feats[static_cast<size_t>(t) * dim + k];
      if (var_norm) x = x * scale_f;
      feats[static_cast<size_t>(t) * dim + k] = x + offset_f;
    }
  }
  return 0;
}

// ---- GMM acoustic model (test infrastructure): DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased
// (gmm/decodable-am-diag-gmm.cc:27-70) with VectorBase<float>::LogSumExp(-1) (matrix/kaldi-vector.cc:760-778), times
// the scale of DecodableAmDiagGmmScaled.  PARITY UNPINNED (no GMM model or features in the tree).
extern "C" void orc_am_gmm_loglikes(int num_pdfs, int dim, const int32_t *mix_off, const float *gconsts, const float *means_invvars,
                                    const float *inv_vars, const float *feats, int T, float scale, float *out) {
  std::vector<float> ll;
  for (int t = 0; t < T; t++) {
    const float *x = feats + static_cast<size_t>(t) * dim;
    for (int p = 0; p < num_pdfs; p++) {
      ll.clear();
      float mx = -INFINITY;
      for (int g = mix_off[p]; g < mix_off[p + 1]; g++) {
        float acc = gconsts[g];
        const float *m = means_invvars + static_cast<size_t>(g) * dim, *v = inv_vars + static_cast<size_t>(g) * dim;
        for (int k = 0; k < dim; k++) acc = acc + m[k] * x[k];
        for (int k = 0; k < dim; k++) acc = acc + (-0.5f * v[k]) * (x[k] * x[k]);
        ll.push_back(acc); mx = std::max(mx, acc);
      }
      const float cutoff = mx + logf(1.1920928955078125e-07f);          // kMinLogDiffFloat = Log(FLT_EPSILON)
      double sum = 0.0;
      for (size_t i = 0; i < ll.size(); i++) if (ll[i] >= cutoff) sum += expf(ll[i] - mx);
      out[static_cast<size_t>(t) * num_pdfs + p] = scale * (mx + static_cast<float>(log(sum)));
    }
  }
}

// ---- add-deltas (test infrastructure): DeltaFeatures ctor + Process (feat/feature-functions.cc:118-165). PARITY UNPINNED.
extern "C" void orc_add_deltas(const float *in, int T, int dim, int order, int window, float *out) {
  std::vector<std::vector<float> > sc(order + 1);
  sc[0].assign(1, 1.0f);
  for (int i = 1; i <= order; i++) {
    const std::vector<float> &prev = sc[i - 1];
    std::vector<float> &cur = sc[i];
    const int po = (static_cast<int>(prev.size()) - 1) / 2, co = po + window;
    cur.assign(prev.size() + 2 * window, 0.0f);
    float normalizer = 0.0f;
    for (int j = -window; j <= window; j++) {
      normalizer += j * j;
      for (int k = -po; k <= po; k++) cur[j + k + co] += static_cast<float>(j) * prev[k + po];
    }
    for (size_t k = 0; k < cur.size(); k++) cur[k] *= 1.0f / normalizer;
  }
  const int n = (order + 1) * dim;
  for (int t = 0; t < T; t++)
    for (int i = 0; i <= order; i++) {
      const int mo = (static_cast<int>(sc[i].size()) - 1) / 2;
      for (int k = 0; k < dim; k++) {
        float acc = 0.f;
        for (int j = -mo; j <= mo; j++) {
          int f = t + j;
          f = f < 0 ? 0 : (f >= T ? T - 1 : f);
          const float s = sc[i][j + mo];
          if (s != 0.0f) acc = acc + s * in[static_cast<size_t>(f) * dim + k];
        }
        out[static_cast<size_t>(t) * n + i * dim + k] = acc;
      }
    }
}
