#define KAMD_RAW_MEMCPY
#include "common.h"

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>

namespace {
// KAMD_ABORT_BACKTRACE=1 (tests/conftest.py sets it): when the process is aborted -- by the HSA runtime after a GPU memory
// fault, by glibc on heap corruption, by std::terminate -- the aborting thread's native stack goes to stderr before the
// default action runs, so that a run that dies says WHO called abort() (GPUTEST_r04.json had only Python's own frames).
struct sigaction g_prev_abort;
void AbortBacktrace(int sig) {
  static const char head[] = "\nkaldi_amd: SIGABRT -- native stack of the aborting thread:\n";
  (void)!write(2, head, sizeof(head) - 1);
  void *frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  sigaction(sig, &g_prev_abort, NULL);      // whoever was there before (Python's faulthandler under pytest) goes next
  raise(sig);
}
struct AbortHook {
  AbortHook() {
    const char *e = getenv("KAMD_ABORT_BACKTRACE");
    if (e && e[0] == '1') {
      void *warm[2];
      (void)backtrace(warm, 2);          // (loads libgcc now: not inside the handler)
      struct sigaction sa;
      memset(&sa, 0, sizeof(sa));
      sa.sa_handler = AbortBacktrace;
      sa.sa_flags = SA_NODEFER | SA_RESETHAND;
      sigaction(SIGABRT, &sa, &g_prev_abort);
    }
  }
} abort_hook;
}  // namespace

namespace kamd {
namespace {
// A small pool of page-locked bounce buffers: a copy takes one, the pool's lock is held only while a buffer changes hands
// (never across a copy or a stream wait: a thread waiting for a long-running stream must not hold up the host-tail threads'
// small reads).  A copy that finds the pool empty allocates a buffer of its own, up to kMaxBounce; beyond that it waits.
constexpr size_t kBounceBytes = 8u << 20;
constexpr int kMaxBounce = 8;
std::mutex g_bounce_mu;
std::condition_variable g_bounce_cv;
void *g_bounce_free[kMaxBounce];
int g_bounce_nfree = 0, g_bounce_made = 0;
struct Bounce {
  void *p = NULL;
  hipError_t err = hipSuccess;
  Bounce() {
    std::unique_lock<std::mutex> lk(g_bounce_mu);
    for (;;) {
      if (g_bounce_nfree > 0) { p = g_bounce_free[--g_bounce_nfree]; return; }
      if (g_bounce_made < kMaxBounce) {
        g_bounce_made++;
        lk.unlock();
        err = ::hipHostMalloc(&p, kBounceBytes, hipHostMallocDefault);      // (never a copy's source or target as far as a caller sees: not noted)
        if (err != hipSuccess) { p = NULL; lk.lock(); g_bounce_made--; g_bounce_cv.notify_one(); }
        return;
      }
      g_bounce_cv.wait(lk);
    }
  }
  ~Bounce() {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(g_bounce_mu); g_bounce_free[g_bounce_nfree++] = p; }
    g_bounce_cv.notify_one();
  }
};
// Which host ranges are page-locked?  The library's OWN list: what it allocated with hipHostMalloc and what it registered
// with hipHostRegister (common.h redirects the four calls here), nothing else.  Until round 6 the runtime was asked
// (hipPointerGetAttributes) -- but the runtime's view goes stale: a range that was registered, or that the runtime pinned for a
// pageable copy and kept in its pin cache, and was then given back to the allocator still reads "hipMemoryTypeHost" when
// the same addresses come back as another, pageable, buffer, and a copy started on that answer faults the GPU on the first page
// that is no longer mapped for it (tools/repro/pinned_neighbour.cc scenario 7: "Memory access fault by GPU node-2 ... on
// address <the mapping>", the abort of rounds 4 / 5 word for word).  A caller's own page-locked memory is therefore treated as
// pageable -- one more memcpy, never a fault.
std::mutex g_pin_mu;
std::map<uintptr_t, size_t> g_pinned;            // base -> bytes
bool HostPinnedRange(const void *p, size_t bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pinned.upper_bound(a);
  if (it == g_pinned.begin()) return false;
  --it;
  return a >= it->first && a + bytes <= it->first + it->second;
}
}  // namespace

void NotePinned(const void *p, size_t bytes) {
  if (!p || !bytes) return;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pinned[reinterpret_cast<uintptr_t>(p)] = bytes;
}
void ForgetPinned(const void *p) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pinned.erase(reinterpret_cast<uintptr_t>(p));
}
bool IsNotedPinned(const void *p, size_t bytes) { return HostPinnedRange(p, bytes); }
hipError_t HostMallocNoted(void **p, size_t bytes, unsigned flags) {
  const hipError_t e = ::hipHostMalloc(p, bytes, flags);
  if (e == hipSuccess) NotePinned(*p, bytes);
  return e;
}
hipError_t HostFreeNoted(void *p) {
  ForgetPinned(p);
  return ::hipHostFree(p);
}
hipError_t HostRegisterNoted(void *p, size_t bytes, unsigned flags) {
  const hipError_t e = ::hipHostRegister(p, bytes, flags);
  if (e == hipSuccess) NotePinned(p, bytes);
  return e;
}
hipError_t HostUnregisterNoted(void *p) {
  ForgetPinned(p);
  return ::hipHostUnregister(p);
}

hipError_t MemcpySafe(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (bytes == 0) return hipSuccess;
  const bool h2d = kind == hipMemcpyHostToDevice && !HostPinnedRange(src, bytes), d2h = kind == hipMemcpyDeviceToHost && !HostPinnedRange(dst, bytes);
  if (!h2d && !d2h) return ::hipMemcpy(dst, src, bytes, kind);
  Bounce bb;
  hipError_t e = bb.err;
  for (size_t off = 0; e == hipSuccess && off < bytes; off += kBounceBytes) {
    const size_t n = std::min(kBounceBytes, bytes - off);
    if (h2d) {
      memcpy(bb.p, static_cast<const char *>(src) + off, n);
      e = ::hipMemcpy(static_cast<char *>(dst) + off, bb.p, n, hipMemcpyHostToDevice);
    } else {
      e = ::hipMemcpy(bb.p, static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost);
      if (e == hipSuccess) memcpy(static_cast<char *>(dst) + off, bb.p, n);
    }
  }
  return e;
}

hipError_t MemcpyAsyncSafe(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  const bool h2d = kind == hipMemcpyHostToDevice && !HostPinnedRange(src, bytes), d2h = kind == hipMemcpyDeviceToHost && !HostPinnedRange(dst, bytes);
  if (!h2d && !d2h) return ::hipMemcpyAsync(dst, src, bytes, kind, st);
  Bounce bb;
  hipError_t e = bb.err;
  for (size_t off = 0; e == hipSuccess && off < bytes; off += kBounceBytes) {
    const size_t n = std::min(kBounceBytes, bytes - off);
    if (h2d) {
      memcpy(bb.p, static_cast<const char *>(src) + off, n);
      e = ::hipMemcpyAsync(static_cast<char *>(dst) + off, bb.p, n, hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);          // (the bounce buffer is reused)
    } else {
      e = ::hipMemcpyAsync(bb.p, static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e == hipSuccess) memcpy(static_cast<char *>(dst) + off, bb.p, n);
    }
  }
  return e;
}

// The pitched copies: rows of `width` bytes, `spitch` / `dpitch` bytes apart.  With a pageable host side the rows are packed
// into (unpacked from) the bounce buffer, as many whole rows per trip as fit.
static hipError_t Memcpy2DImpl(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                               bool async, hipStream_t st) {
  if (width == 0 || height == 0) return hipSuccess;
  const size_t hspan_src = (height - 1) * spitch + width, hspan_dst = (height - 1) * dpitch + width;
  const bool h2d = kind == hipMemcpyHostToDevice && !HostPinnedRange(src, hspan_src), d2h = kind == hipMemcpyDeviceToHost && !HostPinnedRange(dst, hspan_dst);
  if (!h2d && !d2h) return async ? ::hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, st) : ::hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
  if (width > kBounceBytes) {                  // (a row longer than the buffer: row by row through the flat copy)
    hipError_t e = hipSuccess;
    for (size_t r = 0; e == hipSuccess && r < height; r++)
      e = async ? MemcpyAsyncSafe(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width, kind, st)
                : MemcpySafe(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width, kind);
    return e;
  }
  Bounce bb;
  hipError_t e = bb.err;
  const size_t rows_per_trip = kBounceBytes / width;
  for (size_t r0 = 0; e == hipSuccess && r0 < height; r0 += rows_per_trip) {
    const size_t nr = std::min(rows_per_trip, height - r0);
    char *b = static_cast<char *>(bb.p);
    if (h2d) {
      for (size_t r = 0; r < nr; r++) memcpy(b + r * width, static_cast<const char *>(src) + (r0 + r) * spitch, width);
      e = async ? ::hipMemcpy2DAsync(static_cast<char *>(dst) + r0 * dpitch, dpitch, b, width, width, nr, hipMemcpyHostToDevice, st)
                : ::hipMemcpy2D(static_cast<char *>(dst) + r0 * dpitch, dpitch, b, width, width, nr, hipMemcpyHostToDevice);
      if (async && e == hipSuccess) e = hipStreamSynchronize(st);
    } else {
      e = async ? ::hipMemcpy2DAsync(b, width, static_cast<const char *>(src) + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost, st)
                : ::hipMemcpy2D(b, width, static_cast<const char *>(src) + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost);
      if (async && e == hipSuccess) e = hipStreamSynchronize(st);
      if (e == hipSuccess)
        for (size_t r = 0; r < nr; r++) memcpy(static_cast<char *>(dst) + (r0 + r) * dpitch, b + r * width, width);
    }
  }
  return e;
}
hipError_t Memcpy2DSafe(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind) {
  return Memcpy2DImpl(dst, dpitch, src, spitch, width, height, kind, false, NULL);
}
hipError_t Memcpy2DAsyncSafe(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t st) {
  return Memcpy2DImpl(dst, dpitch, src, spitch, width, height, kind, true, st);
}

std::string &LastError() {
  static thread_local std::string s;
  return s;
}
int SetError(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  LastError() = buf;
  return code;
}
bool RequireDevice() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    SetError(KAMD_ERR_HIP, "no HIP device available (%s): kaldi_amd runs on MI355X only, there is no CPU fallback",
             e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    return false;
  }
  return true;
}
}  // namespace kamd

extern "C" {
const char *kamd_last_error(void) { return kamd::LastError().c_str(); }
const char *kamd_version(void) { return "kaldi_amd 0.1 (gfx950)"; }
int kamd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int kamd_device_num_cus(void) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return prop.multiProcessorCount;
}
int kamd_set_device(int device) {
  KAMD_HIP(hipSetDevice(device));
  return KAMD_OK;
}
void *kamd_malloc(size_t bytes) {
  void *p = NULL;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess) { kamd::SetError(KAMD_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return NULL; }
  return p;
}
int kamd_free(void *p) { KAMD_HIP(hipFree(p)); return KAMD_OK; }
// (the caller's memory is pageable as a rule -- a numpy array, a std::vector: through the bounce buffer like every copy of the library)
int kamd_memcpy_h2d(void *d, const void *h, size_t bytes) { KAMD_HIP(kamd::MemcpySafe(d, h, bytes, hipMemcpyHostToDevice)); return KAMD_OK; }
int kamd_memcpy_d2h(void *h, const void *d, size_t bytes) { KAMD_HIP(kamd::MemcpySafe(h, d, bytes, hipMemcpyDeviceToHost)); return KAMD_OK; }
int kamd_device_synchronize(void) { KAMD_HIP(hipDeviceSynchronize()); return KAMD_OK; }
int kamd_device_mem_info(size_t *free_bytes, size_t *total_bytes) { KAMD_HIP(hipMemGetInfo(free_bytes, total_bytes)); return KAMD_OK; }
void kamd_mfcc_opts_default(kamd_mfcc_opts *o) {
  // feat/feature-window.h:54-66, feat/feature-mfcc.h:50-58 (dither forced to 0)
  kamd_frame_opts f = {16000.0f, 10.0f, 25.0f, 0.0f, 0.97f, 1, KAMD_WIN_POVEY, 1, 0.42f, 1};
  kamd_mel_opts m = {23, 20.0f, 0.0f, 100.0f, -500.0f, 0};
  o->frame = f; o->mel = m;
  o->num_ceps = 13; o->use_energy = 1; o->energy_floor = 0.0f; o->raw_energy = 1;
  o->cepstral_lifter = 22.0f; o->htk_compat = 0;
}
void kamd_fbank_opts_default(kamd_fbank_opts *o) {
  kamd_frame_opts f = {16000.0f, 10.0f, 25.0f, 0.0f, 0.97f, 1, KAMD_WIN_POVEY, 1, 0.42f, 1};
  kamd_mel_opts m = {23, 20.0f, 0.0f, 100.0f, -500.0f, 0};
  o->frame = f; o->mel = m;
  o->use_energy = 0; o->energy_floor = 0.0f; o->raw_energy = 1; o->htk_compat = 0;
  o->use_log_fbank = 1; o->use_power = 1;
}
void kamd_decoder_config_default(kamd_decoder_config *c) {
  // decoder/lattice-faster-decoder.h:56-64
  c->beam = 16.0f; c->max_active = 2147483647; c->min_active = 200; c->lattice_beam = 10.0f;
  c->prune_interval = 25; c->beam_delta = 0.5f; c->hash_ratio = 2.0f; c->prune_scale = 0.1f;
}
void kamd_decoder_sizes_default(kamd_decoder_sizes *s) {
  s->max_lanes = 64; s->hash_capacity = 1 << 17; s->arena_tokens = 1 << 22;
  s->arena_links = 1 << 23; s->max_frames = 2048;
}
// Arenas for `max_lanes` lanes of at most `max_frames` decoded frames each, `avg_frames` (0: max_frames) on average
// (the pools are split between lanes in proportion to utterance length, kamd_decoder_reserve), kept within
// `hbm_fraction` of the HBM that is free now (16 B per token record incl. its map entry, 24 B per link).
int kamd_decoder_sizes_suggest(const kamd_decoder_config *cfg, int max_lanes, int max_frames, int avg_frames, int hash_capacity,
                               int tokens_per_frame, int links_per_frame, float hbm_fraction, kamd_decoder_sizes *out) {
  if (!cfg || !out || max_lanes < 1 || max_frames < 1) return kamd::SetError(KAMD_ERR_ARG, "kamd_decoder_sizes_suggest: bad arguments");
  const double act = cfg->max_active < INT32_MAX ? cfg->max_active : 20000;
  int64_t tpf = tokens_per_frame > 0 ? tokens_per_frame : static_cast<int64_t>(std::min(3.0 * act, 60000.0) + 2000);
  int64_t lpf = links_per_frame > 0 ? links_per_frame : static_cast<int64_t>(1.6 * tpf + 2000);
  int64_t hc = hash_capacity;
  if (hc <= 0) {
    const int64_t want = std::max<int64_t>(std::max<int64_t>(4 * tpf, max_frames + 8), 4096);
    hc = 1;
    while (hc < want) hc <<= 1;
  }
  const int64_t avg = (avg_frames > 0 ? avg_frames : max_frames) + 2;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
    const double need = static_cast<double>(max_lanes) * avg * (16.0 * tpf + 24.0 * lpf);
    const double budget = static_cast<double>(hbm_fraction > 0 ? hbm_fraction : 0.5f) * free_b;
    if (need > budget) {
      const double k = budget / need;
      tpf = std::max<int64_t>(4000, static_cast<int64_t>(tpf * k));
      lpf = std::max<int64_t>(6000, static_cast<int64_t>(lpf * k));
    }
  } else {
    (void)hipGetLastError();
  }
  out->max_lanes = max_lanes; out->hash_capacity = static_cast<int32_t>(hc);
  out->arena_tokens = tpf * avg; out->arena_links = lpf * avg; out->max_frames = max_frames + 1;
  return KAMD_OK;
}
}
