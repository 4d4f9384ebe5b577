// feat.hip -- MFCC / fbank Compute() on device (gfx950).
//
// Replaces, per frame: ExtractWindow + ProcessWindow (feat/feature-window.cc:133-220),
// SplitRadixRealFft::Compute (matrix/srfft.cc:356) + ComputePowerSpectrum
// (feat/feature-functions.cc:29-51), MelBanks::Compute (feat/mel-computations.cc:
// 226-252), and MfccComputer::Compute / FbankComputer::Compute
// (feat/feature-mfcc.cc:28-80, feat/feature-fbank.cc:73-122).
//
// Layout: one 64-lane wavefront owns one frame; 4 frames per 256-thread workgroup.
// The frame lives in LDS (re/im, 2*N floats per wave); the N-point FFT is an
// in-LDS radix-2 DIT with a host-computed twiddle table; mel / DCT are one lane per
// output bin.  Bytes per frame: 1.6 kB in, <=160 B out -> HBM-trivial; the kernel
// exists so that features never cross PCIe.
#include <cmath>
#include <vector>

#include "common.h"

namespace kamd {

struct FeatDev {
  // options
  int frame_len, frame_shift, N, log2N;
  int snip_edges, remove_dc, htk_mode_floor;
  float preemph;
  int num_bins, num_out;  // num_out = feature dim
  int is_mfcc, num_ceps;
  int use_energy, raw_energy, htk_compat, use_log, use_power;
  float energy_floor_log;
  int has_energy_floor, has_lifter;
  // tables (device)
  const float *window;    // [frame_len]
  const float *tw_cos;    // [N/2]
  const float *tw_sin;    // [N/2]  (negative sine: forward transform)
  const int *mel_first;   // [num_bins]
  const int *mel_len;     // [num_bins]
  const int *mel_off;     // [num_bins] offset into mel_w
  const float *mel_w;     // packed weights
  const float *dct;       // [num_ceps x num_bins]
  const float *lifter;    // [num_ceps]
};

__device__ inline float WaveSum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// utt lookup: largest u with frame_off[u] <= g
__device__ inline int FindUtt(const int64_t *off, int n, int64_t g) {
  int lo = 0, hi = n;  // off has n+1 entries
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// A wavefront works on its own frame in its own LDS region: ordering its LDS accesses needs no workgroup barrier (the
// four frames of a workgroup would only wait for each other, a dozen times per frame), just that the compiler keeps the
// order -- one wavefront's LDS instructions execute in issue order.
__device__ inline void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void FeatKernel(
    FeatDev fd, const float *__restrict__ waves, const int64_t *__restrict__ wave_off,
    const int64_t *__restrict__ frame_off /* [n_utts+1] cumulative frames */,
    const int64_t *__restrict__ row_off /* [n_utts] output row of frame 0 */, int n_utts,
    float *__restrict__ out, int ld_out, int frame0 /* index of the first frame (streaming) */,
    const int *__restrict__ frame0_per_utt /* or NULL: per-utterance first frame (batched streaming) */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int N = fd.N;
  float *re = smem + wave * (2 * N + 128);
  float *im = re + N;
  float *aux = im + N;  // [128]: log-mel energies etc.
  // twiddles of the whole workgroup, staged once (every butterfly of every stage reads a pair)
  float *tw_c = smem + 4 * (2 * N + 128), *tw_s = tw_c + (N >> 1);
  for (int i = threadIdx.x; i < (N >> 1); i += 256) { tw_c[i] = fd.tw_cos[i]; tw_s[i] = fd.tw_sin[i]; }
  __syncthreads();
  const bool pairs = n_utts < 0;          // wave_off holds (start, end) pairs: items are not adjacent
  if (pairs) n_utts = -n_utts;
  const int64_t total = frame_off[n_utts];
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const bool live = g < total;
  int u = 0, f = 0;
  int64_t nsamp = 1;
  const float *wav = waves;
  if (live) {
    u = FindUtt(frame_off, n_utts, g);
    if (frame0_per_utt) frame0 = frame0_per_utt[u];
    f = static_cast<int>(g - frame_off[u]) + frame0;
    wav = waves + (pairs ? wave_off[2 * u] : wave_off[u]);
    nsamp = pairs ? wave_off[2 * u + 1] - wave_off[2 * u] : wave_off[u + 1] - wave_off[u];
  }
  // --- ExtractWindow: FirstSampleOfFrame + reflection (feature-window.cc:28-39,191-208)
  int64_t start;
  if (fd.snip_edges) start = static_cast<int64_t>(f) * fd.frame_shift;
  else start = static_cast<int64_t>(fd.frame_shift) * f + fd.frame_shift / 2 - fd.frame_len / 2;
  const int len = fd.frame_len;
  // each lane owns samples lane, lane+64, ... (<= 16 per lane for N <= 1024)
  float x[16];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    int s = lane + 64 * j;
    float v = 0.f;
    if (live && s < len) {
      int64_t i = s + start;
      while (i < 0 || i >= nsamp) {
        if (i < 0) i = -i - 1; else i = 2 * nsamp - 1 - i;
      }
      v = wav[i];
    }
    x[j] = v;
    sum += v;
  }
  // --- ProcessWindow: remove DC, raw log energy, pre-emphasis, window (:133-156)
  if (fd.remove_dc) {
    float mean = WaveSum(sum) / len;
#pragma unroll
    for (int j = 0; j < 16; j++) if (lane + 64 * j < len) x[j] -= mean;
  }
  float raw_log_energy = 0.f;
  if (fd.use_energy && fd.raw_energy) {
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) e += x[j] * x[j];
    e = WaveSum(e);
    raw_log_energy = logf(fmaxf(e, 1.1920928955078125e-07f));  // FLT_EPSILON
  }
  // stage un-windowed samples in LDS so every lane can read its left neighbour
#pragma unroll
  for (int j = 0; j < 16; j++) { int s = lane + 64 * j; if (s < N) re[s] = x[j]; }
  WaveSync();
  if (fd.preemph != 0.f) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      int s = lane + 64 * j;
      if (s < len) {
        float prev = re[s > 0 ? s - 1 : 0];
        x[j] = x[j] - fd.preemph * prev;   // Preemphasize (:100-107)
      }
    }
  }
  WaveSync();
  float win_energy = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    int s = lane + 64 * j;
    if (s < len) x[j] *= fd.window[s];
    win_energy += (s < len) ? x[j] * x[j] : 0.f;
  }
  float signal_log_energy = raw_log_energy;
  if (fd.use_energy && !fd.raw_energy) {   // feature-mfcc.cc:37-39
    float e = WaveSum(win_energy);
    signal_log_energy = logf(fmaxf(e, 1.17549435e-38f));  // FLT_MIN
  }
  // --- FFT: bit-reversed scatter then log2N radix-2 DIT stages in LDS
  const int lg = fd.log2N;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    int s = lane + 64 * j;
    if (s < N) {
      int r = __brev(static_cast<unsigned>(s)) >> (32 - lg);
      re[r] = (s < len) ? x[j] : 0.f;
      im[r] = 0.f;
    }
  }
  WaveSync();
  const int half_n = N >> 1;
  for (int st = 0; st < lg; st++) {
    const int half = 1 << st;
    for (int b = lane; b < half_n; b += 64) {
      int k = b & (half - 1);
      int i0 = ((b >> st) << (st + 1)) + k;
      int i1 = i0 + half;
      int tw = k << (lg - 1 - st);
      float wr = tw_c[tw], wi = tw_s[tw];
      float xr = re[i1], xi = im[i1];
      float tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
      float ar = re[i0], ai = im[i0];
      re[i1] = ar - tr; im[i1] = ai - ti;
      re[i0] = ar + tr; im[i0] = ai + ti;
    }
    WaveSync();
  }
  // --- power spectrum bins 0..N/2 (feature-functions.cc:29-51), kept in re[]
  for (int k = lane; k <= half_n; k += 64) {
    float p = re[k] * re[k] + im[k] * im[k];
    if (!fd.use_power) p = sqrtf(p);   // FbankComputer: ApplyPow(0.5) (feature-fbank.cc:97-98)
    im[k] = p;                         // write to im[] to avoid racing with re[] readers
  }
  WaveSync();
  // --- mel filterbank: one lane per bin (mel-computations.cc:226-252)
  for (int b = lane; b < fd.num_bins; b += 64) {
    const float *w = fd.mel_w + fd.mel_off[b];
    const int first = fd.mel_first[b], n = fd.mel_len[b];
    float e = 0.f;
    for (int j = 0; j < n; j++) e += w[j] * im[first + j];
    if (fd.htk_mode_floor && e < 1.0f) e = 1.0f;
    if (fd.use_log) e = logf(fmaxf(e, 1.1920928955078125e-07f));
    aux[b] = e;
  }
  WaveSync();
  if (!live) return;
  float *orow = out + (row_off[u] + (f - frame0)) * static_cast<int64_t>(ld_out);
  if (fd.is_mfcc) {
    // feature-mfcc.cc:56-79: DCT, lifter, energy / C0, htk_compat reorder
    const int C = fd.num_ceps;
    if (fd.use_energy && fd.has_energy_floor && signal_log_energy < fd.energy_floor_log)
      signal_log_energy = fd.energy_floor_log;
    for (int c = lane; c < C; c += 64) {
      const float *d = fd.dct + c * fd.num_bins;
      float s = 0.f;
      for (int m = 0; m < fd.num_bins; m++) s += d[m] * aux[m];
      if (fd.has_lifter) s *= fd.lifter[c];
      if (fd.use_energy && c == 0) s = signal_log_energy;
      int oc = c;
      if (fd.htk_compat) {
        if (c == 0) { oc = C - 1; if (!fd.use_energy) s *= 1.41421356237309504880f; }
        else oc = c - 1;
      }
      orow[oc] = s;
    }
    for (int c = C + lane; c < ld_out; c += 64) orow[c] = 0.f;
  } else {
    // feature-fbank.cc:100-121
    const int nb = fd.num_bins;
    const int mel_offset = (fd.use_energy && !fd.htk_compat) ? 1 : 0;
    for (int b = lane; b < nb; b += 64) orow[mel_offset + b] = aux[b];
    if (fd.use_energy && lane == 0) {
      if (fd.has_energy_floor && signal_log_energy < fd.energy_floor_log)
        signal_log_energy = fd.energy_floor_log;
      orow[fd.htk_compat ? nb : 0] = signal_log_energy;
    }
    for (int c = fd.num_out + lane; c < ld_out; c += 64) orow[c] = 0.f;
  }
}

// ---------------------------------------------------------------- host side
struct Feat {
  FeatDev dev;
  kamd_frame_opts fo;
  std::vector<void *> allocs;
  // scratch for the host-convenience entry point
  float *d_wave = NULL, *d_out = NULL;
  int64_t *d_meta = NULL;
  size_t wave_cap = 0, out_cap = 0, meta_cap = 0;
};

inline float MelScale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
inline float InverseMelScale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

// MelBanks::VtlnWarpFreq (feat/mel-computations.cc:150-211)
static float VtlnWarpFreq(float vl, float vh, float low, float high, float warp, float freq) {
  if (freq < low || freq > high) return freq;
  float l = vl * fmaxf(1.0f, warp), h = vh * fminf(1.0f, warp);
  float scale = 1.0f / warp, Fl = scale * l, Fh = scale * h;
  float sl = (Fl - low) / (l - low), sr = (high - Fh) / (high - h);
  if (freq < l) return low + sl * (freq - low);
  if (freq < h) return scale * freq;
  return high + sr * (freq - high);
}

template <typename T>
static const T *Upload(Feat *f, const std::vector<T> &v) {
  T *d = DevAlloc<T>(v.size());
  if (!d) return NULL;
  f->allocs.push_back(d);
  if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
    return NULL;
  return d;
}

static Feat *CreateCommon(const kamd_frame_opts &fo, const kamd_mel_opts &mo, float warp,
                          bool is_mfcc, int num_ceps, float lifter_q) {
  if (fo.dither != 0.0f) {
    SetError(KAMD_ERR_ARG, "device features need dither=0 (the reference's dither is a "
             "host RNG, feat/feature-window.cc:90-98)");
    return NULL;
  }
  int len = static_cast<int>(fo.samp_freq * 0.001f * fo.frame_length_ms);
  int shift = static_cast<int>(fo.samp_freq * 0.001f * fo.frame_shift_ms);
  int N = 1;
  while (N < len) N <<= 1;
  if (!fo.round_to_power_of_two && N != len) {
    SetError(KAMD_ERR_ARG, "device FFT needs a power-of-two window (round_to_power_of_two)");
    return NULL;
  }
  if (N > 1024 || N < 128 || len < 2 || shift < 1 || mo.num_bins < 3 || mo.num_bins > 128) {
    SetError(KAMD_ERR_ARG, "unsupported frame options (padded window %d, bins %d)", N, mo.num_bins);
    return NULL;
  }
  if (!RequireDevice()) return NULL;
  Feat *f = new Feat();
  f->fo = fo;
  FeatDev &d = f->dev;
  memset(&d, 0, sizeof(d));
  d.frame_len = len; d.frame_shift = shift; d.N = N;
  d.log2N = 0; while ((1 << d.log2N) < N) d.log2N++;
  d.snip_edges = fo.snip_edges; d.remove_dc = fo.remove_dc_offset; d.preemph = fo.preemph_coeff;
  d.num_bins = mo.num_bins; d.htk_mode_floor = mo.htk_mode;
  // FeatureWindowFunction (feature-window.cc:109-131): double math, float storage
  std::vector<float> win(len);
  double a = 2.0 * M_PI / (len - 1);
  for (int i = 0; i < len; i++) {
    double x = i, w;
    switch (fo.window_type) {
      case KAMD_WIN_HANNING: w = 0.5 - 0.5 * cos(a * x); break;
      case KAMD_WIN_HAMMING: w = 0.54 - 0.46 * cos(a * x); break;
      case KAMD_WIN_POVEY: w = pow(0.5 - 0.5 * cos(a * x), 0.85); break;
      case KAMD_WIN_RECTANGULAR: w = 1.0; break;
      default: w = fo.blackman_coeff - 0.5 * cos(a * x) + (0.5 - fo.blackman_coeff) * cos(2 * a * x);
    }
    win[i] = static_cast<float>(w);
  }
  std::vector<float> tc(N / 2), ts(N / 2);
  for (int k = 0; k < N / 2; k++) {
    double ang = -2.0 * M_PI * k / N;
    tc[k] = static_cast<float>(cos(ang));
    ts[k] = static_cast<float>(sin(ang));
  }
  // MelBanks::MelBanks (mel-computations.cc:33-133)
  float nyquist = 0.5f * fo.samp_freq;
  float low = mo.low_freq, high = mo.high_freq > 0.0f ? mo.high_freq : nyquist + mo.high_freq;
  if (low < 0.0f || low >= nyquist || high <= 0.0f || high > nyquist || high <= low) {
    SetError(KAMD_ERR_ARG, "bad mel options: low-freq %f high-freq %f nyquist %f", low, high, nyquist);
    delete f;
    return NULL;
  }
  float bin_width = fo.samp_freq / N;
  float mel_low = MelScale(low), mel_high = MelScale(high);
  float delta = (mel_high - mel_low) / (mo.num_bins + 1);
  float vl = mo.vtln_low, vh = mo.vtln_high;
  if (vh < 0.0f) vh += nyquist;
  std::vector<int> first(mo.num_bins), mlen(mo.num_bins), moff(mo.num_bins);
  std::vector<float> mw;
  for (int b = 0; b < mo.num_bins; b++) {
    float left = mel_low + b * delta, center = mel_low + (b + 1) * delta,
          right = mel_low + (b + 2) * delta;
    if (warp != 1.0f) {
      left = MelScale(VtlnWarpFreq(vl, vh, low, high, warp, InverseMelScale(left)));
      center = MelScale(VtlnWarpFreq(vl, vh, low, high, warp, InverseMelScale(center)));
      right = MelScale(VtlnWarpFreq(vl, vh, low, high, warp, InverseMelScale(right)));
    }
    int fi = -1, li = -1;
    std::vector<float> wts(N / 2, 0.0f);
    for (int i = 0; i < N / 2; i++) {
      float mel = MelScale(bin_width * i);
      if (mel > left && mel < right) {
        wts[i] = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
        if (fi == -1) fi = i;
        li = i;
      }
    }
    if (fi == -1) {
      SetError(KAMD_ERR_ARG, "empty mel bin %d (num-mel-bins too large)", b);
      delete f;
      return NULL;
    }
    first[b] = fi; mlen[b] = li + 1 - fi; moff[b] = static_cast<int>(mw.size());
    for (int i = fi; i <= li; i++) mw.push_back(wts[i]);
    if (mo.htk_mode && b == 0 && mel_low != 0.0f) mw[moff[b]] = 0.0f;  // :121-123
  }
  d.window = Upload(f, win); d.tw_cos = Upload(f, tc); d.tw_sin = Upload(f, ts);
  d.mel_first = Upload(f, first); d.mel_len = Upload(f, mlen); d.mel_off = Upload(f, moff);
  d.mel_w = Upload(f, mw);
  d.is_mfcc = is_mfcc;
  if (is_mfcc) {
    // ComputeDctMatrix (matrix/matrix-functions.cc:592-608), first num_ceps rows
    int nb = mo.num_bins;
    std::vector<float> dct(num_ceps * nb), lif(num_ceps, 1.0f);
    float n0 = std::sqrt(1.0 / static_cast<float>(nb)), n1 = std::sqrt(2.0 / static_cast<float>(nb));
    for (int j = 0; j < nb; j++) dct[j] = n0;
    for (int k = 1; k < num_ceps; k++)
      for (int n = 0; n < nb; n++)
        dct[k * nb + n] = n1 * std::cos(static_cast<double>(M_PI) / nb * (n + 0.5) * k);
    if (lifter_q != 0.0f)  // ComputeLifterCoeffs (mel-computations.cc:253-259)
      for (int i = 0; i < num_ceps; i++) lif[i] = 1.0 + 0.5 * lifter_q * sin(M_PI * i / lifter_q);
    d.dct = Upload(f, dct); d.lifter = Upload(f, lif);
    d.has_lifter = lifter_q != 0.0f;
    d.num_ceps = num_ceps;
    if (!d.dct || !d.lifter) { SetError(KAMD_ERR_HIP, "feature table upload failed"); delete f; return NULL; }
  }
  if (!d.window || !d.tw_cos || !d.tw_sin || !d.mel_first || !d.mel_len || !d.mel_off || !d.mel_w) {
    SetError(KAMD_ERR_HIP, "feature table upload failed");
    delete f;
    return NULL;
  }
  return f;
}

}  // namespace kamd

using kamd::Feat;

extern "C" {

kamd_feat *kamd_mfcc_create(const kamd_mfcc_opts *o, float vtln_warp) {
  if (o->num_ceps > o->mel.num_bins || o->num_ceps < 1) {
    kamd::SetError(KAMD_ERR_ARG, "num-ceps cannot be larger than num-mel-bins");  // feature-mfcc.cc:87-91
    return NULL;
  }
  Feat *f = kamd::CreateCommon(o->frame, o->mel, vtln_warp, true, o->num_ceps, o->cepstral_lifter);
  if (!f) return NULL;
  kamd::FeatDev &d = f->dev;
  d.num_out = o->num_ceps;
  d.use_energy = o->use_energy; d.raw_energy = o->raw_energy; d.htk_compat = o->htk_compat;
  d.use_log = 1; d.use_power = 1;
  d.has_energy_floor = o->energy_floor > 0.0f;
  d.energy_floor_log = d.has_energy_floor ? logf(o->energy_floor) : 0.0f;
  return reinterpret_cast<kamd_feat *>(f);
}

kamd_feat *kamd_fbank_create(const kamd_fbank_opts *o, float vtln_warp) {
  Feat *f = kamd::CreateCommon(o->frame, o->mel, vtln_warp, false, 0, 0.0f);
  if (!f) return NULL;
  kamd::FeatDev &d = f->dev;
  d.num_out = o->mel.num_bins + (o->use_energy ? 1 : 0);
  d.use_energy = o->use_energy; d.raw_energy = o->raw_energy; d.htk_compat = o->htk_compat;
  d.use_log = o->use_log_fbank; d.use_power = o->use_power;
  d.has_energy_floor = o->energy_floor > 0.0f;
  d.energy_floor_log = d.has_energy_floor ? logf(o->energy_floor) : 0.0f;
  return reinterpret_cast<kamd_feat *>(f);
}

void kamd_feat_destroy(kamd_feat *h) {
  Feat *f = reinterpret_cast<Feat *>(h);
  if (!f) return;
  for (size_t i = 0; i < f->allocs.size(); i++) (void)hipFree(f->allocs[i]);
  if (f->d_wave) (void)hipFree(f->d_wave);
  if (f->d_out) (void)hipFree(f->d_out);
  if (f->d_meta) (void)hipFree(f->d_meta);
  delete f;
}

int kamd_feat_dim(const kamd_feat *h) { return reinterpret_cast<const Feat *>(h)->dev.num_out; }

int kamd_feat_num_frames(const kamd_feat *h, int64_t num_samples) {
  // NumFrames(..., flush=true) (feat/feature-window.cc:41-87)
  const kamd::FeatDev &d = reinterpret_cast<const Feat *>(h)->dev;
  int64_t shift = d.frame_shift, len = d.frame_len;
  if (d.snip_edges) {
    if (num_samples < len) return 0;
    return static_cast<int>(1 + (num_samples - len) / shift);
  }
  return static_cast<int>((num_samples + shift / 2) / shift);
}

int kamd_feat_compute_batch_device(kamd_feat *h, const float *d_waves, const int64_t *h_wave_off,
                                   int n_utts, float *d_out, const int64_t *h_row_off, int ld_out,
                                   void *stream) {
  Feat *f = reinterpret_cast<Feat *>(h);
  if (n_utts <= 0) return KAMD_OK;
  if (ld_out < f->dev.num_out) return kamd::SetError(KAMD_ERR_ARG, "ld_out < feature dim");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // meta: wave_off[n+1] | frame_off[n+1] | row_off[n]
  std::vector<int64_t> meta(3 * (n_utts + 1));
  int64_t tot = 0;
  for (int u = 0; u < n_utts; u++) {
    meta[u] = h_wave_off[u];
    meta[(n_utts + 1) + u] = tot;
    tot += kamd_feat_num_frames(h, h_wave_off[u + 1] - h_wave_off[u]);
    meta[2 * (n_utts + 1) + u] = h_row_off[u];
  }
  meta[n_utts] = h_wave_off[n_utts];
  meta[(n_utts + 1) + n_utts] = tot;
  if (tot == 0) return KAMD_OK;
  if (meta.size() > f->meta_cap) {
    KAMD_HIP(hipStreamSynchronize(st));
    if (f->d_meta) KAMD_HIP(hipFree(f->d_meta));
    f->d_meta = NULL; f->meta_cap = 0;
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_meta), 2 * meta.size() * 8));
    f->meta_cap = 2 * meta.size();
  }
  int64_t *d_meta = f->d_meta;
  KAMD_HIP(hipMemcpyAsync(d_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));  // 'meta' is a host temporary
  int blocks = kamd::CeilDiv(tot, 4);
  size_t lds = (4 * (2 * f->dev.N + 128) + f->dev.N) * sizeof(float);
  hipLaunchKernelGGL(kamd::FeatKernel, dim3(blocks), dim3(256), lds, st, f->dev, d_waves, d_meta,
                     d_meta + (n_utts + 1), d_meta + 2 * (n_utts + 1), n_utts, d_out, ld_out, 0, NULL);
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}

}  // extern "C"
namespace kamd {
int FeatBuildMeta(kamd_feat *h, const int64_t *h_wave_off, int n_utts, const int64_t *h_row_off, int64_t *meta, int64_t *total_frames) {
  int64_t tot = 0;
  for (int u = 0; u < n_utts; u++) {
    meta[u] = h_wave_off[u];
    meta[(n_utts + 1) + u] = tot;
    tot += kamd_feat_num_frames(h, h_wave_off[u + 1] - h_wave_off[u]);
    meta[2 * (n_utts + 1) + u] = h_row_off[u];
  }
  meta[n_utts] = h_wave_off[n_utts];
  meta[(n_utts + 1) + n_utts] = tot;
  meta[2 * (n_utts + 1) + n_utts] = 0;
  *total_frames = tot;
  return KAMD_OK;
}
int FeatLaunchPremeta(kamd_feat *h, const float *d_waves, const int64_t *d_meta, int n_utts, int64_t total_frames, float *d_out,
                      int ld_out, hipStream_t st) {
  Feat *f = reinterpret_cast<Feat *>(h);
  if (n_utts <= 0 || total_frames <= 0) return KAMD_OK;
  if (ld_out < f->dev.num_out) return SetError(KAMD_ERR_ARG, "ld_out < feature dim");
  const int blocks = CeilDiv(total_frames, 4);
  const size_t lds = (4 * (2 * f->dev.N + 128) + f->dev.N) * sizeof(float);
  hipLaunchKernelGGL(FeatKernel, dim3(blocks), dim3(256), lds, st, f->dev, d_waves, d_meta, d_meta + (n_utts + 1),
                     d_meta + 2 * (n_utts + 1), n_utts, d_out, ld_out, 0, NULL);
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}
}  // namespace kamd
extern "C" {

// Frames [first_frame[u], first_frame[u] + num_frames[u]) of n device-resident waveforms in
// ONE launch (batched streaming: the frames that became computable on every stream).
int kamd_feat_compute_ranges_device(kamd_feat *h, const float *d_waves, const int64_t *h_wave_start,
                                    const int64_t *h_wave_len, const int32_t *h_first_frame,
                                    const int32_t *h_num_frames, int n, float *d_out, const int64_t *h_row_off,
                                    int ld_out, void *stream) {
  Feat *f = reinterpret_cast<Feat *>(h);
  if (n <= 0) return KAMD_OK;
  if (ld_out < f->dev.num_out) return kamd::SetError(KAMD_ERR_ARG, "ld_out < feature dim");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // meta: (wave start, wave end) pairs [2(n+1)] | frame_off[n+1] | row_off[n+1] | first_frame[n] (int32);
  // n_utts is passed negated to tell the kernel that the wave offsets are pairs
  std::vector<int64_t> meta(4 * (n + 1) + (n + 1) / 2 + 1, 0);
  int64_t tot = 0;
  int32_t *f0 = reinterpret_cast<int32_t *>(&meta[4 * (n + 1)]);
  for (int u = 0; u < n; u++) {
    meta[2 * u] = h_wave_start[u]; meta[2 * u + 1] = h_wave_start[u] + h_wave_len[u];
    meta[2 * (n + 1) + u] = tot;
    tot += h_num_frames[u] > 0 ? h_num_frames[u] : 0;
    meta[3 * (n + 1) + u] = h_row_off[u];
    f0[u] = h_first_frame[u];
  }
  meta[2 * (n + 1) + n] = tot;
  if (tot == 0) return KAMD_OK;
  if (meta.size() > f->meta_cap) {
    KAMD_HIP(hipStreamSynchronize(st));
    if (f->d_meta) KAMD_HIP(hipFree(f->d_meta));
    f->d_meta = NULL; f->meta_cap = 0;
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_meta), 2 * meta.size() * 8));
    f->meta_cap = 2 * meta.size();
  }
  int64_t *d_meta = f->d_meta;
  KAMD_HIP(hipMemcpyAsync(d_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));  // 'meta' is a host temporary
  const size_t lds = (4 * (2 * f->dev.N + 128) + f->dev.N) * sizeof(float);
  hipLaunchKernelGGL(kamd::FeatKernel, dim3(kamd::CeilDiv(tot, 4)), dim3(256), lds, st, f->dev, d_waves, d_meta,
                     d_meta + 2 * (n + 1), d_meta + 3 * (n + 1), -n, d_out, ld_out, 0,
                     reinterpret_cast<const int *>(d_meta + 4 * (n + 1)));
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}

int kamd_feat_num_frames_flush(const kamd_feat *h, int64_t num_samples, int flush) {
  // NumFrames(num_samples, opts, flush) (feat/feature-window.cc:41-87)
  const kamd::FeatDev &d = reinterpret_cast<const Feat *>(h)->dev;
  const int64_t shift = d.frame_shift, len = d.frame_len;
  if (d.snip_edges) return num_samples < len ? 0 : static_cast<int>(1 + (num_samples - len) / shift);
  int num_frames = static_cast<int>((num_samples + shift / 2) / shift);
  if (flush) return num_frames;
  int64_t end_of_last = (shift * (num_frames - 1) + shift / 2 - len / 2) + len;
  while (num_frames > 0 && end_of_last > num_samples) { num_frames--; end_of_last -= shift; }
  return num_frames;
}

// frames [first_frame, first_frame + num_frames) of ONE waveform of num_samples samples
// (streaming: the waveform grows, earlier frames were computed before)
int kamd_feat_compute_frames_device(kamd_feat *h, const float *d_wave, int64_t num_samples, int first_frame,
                                    int num_frames, float *d_out, int ld_out, void *stream) {
  Feat *f = reinterpret_cast<Feat *>(h);
  if (num_frames <= 0) return KAMD_OK;
  if (ld_out < f->dev.num_out) return kamd::SetError(KAMD_ERR_ARG, "ld_out < feature dim");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t meta[5] = {0, num_samples, 0, num_frames, 0};   // wave_off[2] | frame_off[2] | row_off[1]
  if (f->meta_cap < 8) {
    if (f->d_meta) KAMD_HIP(hipFree(f->d_meta));
    f->d_meta = NULL; f->meta_cap = 0;
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&f->d_meta), 64 * 8));
    f->meta_cap = 64;
  }
  KAMD_HIP(hipMemcpyAsync(f->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  const size_t lds = (4 * (2 * f->dev.N + 128) + f->dev.N) * sizeof(float);
  hipLaunchKernelGGL(kamd::FeatKernel, dim3(kamd::CeilDiv(num_frames, 4)), dim3(256), lds, st, f->dev, d_wave, f->d_meta,
                     f->d_meta + 2, f->d_meta + 4, 1, d_out, ld_out, first_frame, NULL);
  KAMD_HIP(hipGetLastError());
  return KAMD_OK;
}

int kamd_feat_compute(kamd_feat *h, const float *wave, int64_t num_samples, float *out,
                      int out_rows_cap) {
  Feat *f = reinterpret_cast<Feat *>(h);
  int T = kamd_feat_num_frames(h, num_samples);
  if (T > out_rows_cap) return kamd::SetError(KAMD_ERR_ARG, "output buffer too small");
  if (T == 0) return 0;
  int dim = f->dev.num_out;
  float *d_wave = NULL, *d_out = NULL;
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_wave), num_samples * sizeof(float)));
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&d_out), static_cast<size_t>(T) * dim * sizeof(float)));
  KAMD_HIP(hipMemcpy(d_wave, wave, num_samples * sizeof(float), hipMemcpyHostToDevice));
  int64_t woff[2] = {0, num_samples}, roff[1] = {0};
  int rc = kamd_feat_compute_batch_device(h, d_wave, woff, 1, d_out, roff, dim, NULL);
  if (rc == KAMD_OK) {
    hipError_t e = hipMemcpy(out, d_out, static_cast<size_t>(T) * dim * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = kamd::SetError(KAMD_ERR_HIP, "D2H failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_wave);
  (void)hipFree(d_out);
  return rc == KAMD_OK ? T : rc;
}

}  // extern "C"
