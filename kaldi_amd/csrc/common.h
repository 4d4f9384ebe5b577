// common.h -- shared host/device helpers for the MI355X (gfx950) library.
#ifndef KAMD_COMMON_H_
#define KAMD_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/kaldi_amd.h"

namespace kamd {

// thread-local error string behind kamd_last_error() (a C-ABI must not throw).
std::string &LastError();
int SetError(int code, const char *fmt, ...);
// true iff a HIP device is usable; otherwise sets the error (no CPU fallback exists).
bool RequireDevice();
// rows [src_row[i], +count[i]) of src -> rows [dst_row[i], ...) of dst (descriptors on the device); nnet.hip
int CopyRowBlocks(const float *src, int ld_src, float *dst, int ld_dst, const int64_t *d_src_row, const int64_t *d_dst_row,
                  const int *d_count, int n_items, int max_count, int cols, hipStream_t st);

// feat.hip: the batch launch of kamd_feat_compute_batch_device in two halves, so that a caller can build and upload the
// offsets of many launches once (no host synchronisation per launch).  meta = wave_off[n+1] | frame_off[n+1] | row_off[n+1].
int FeatBuildMeta(kamd_feat *f, const int64_t *h_wave_off, int n_utts, const int64_t *h_row_off, int64_t *meta, int64_t *total_frames);
int FeatLaunchPremeta(kamd_feat *f, const float *d_waves, const int64_t *d_meta, int n_utts, int64_t total_frames, float *d_out,
                      int ld_out, hipStream_t st);

#define KAMD_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess)                                                           \
      return kamd::SetError(KAMD_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                            hipGetErrorString(e_), __FILE__, __LINE__);             \
  } while (0)

#define KAMD_HIP_NULL(call)                                                         \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      kamd::SetError(KAMD_ERR_HIP, "%s failed: %s (%s:%d)", #call,                 \
                     hipGetErrorString(e_), __FILE__, __LINE__);                    \
      return NULL;                                                                  \
    }                                                                               \
  } while (0)

// Copies between PAGEABLE host memory and the device go through a page-locked bounce buffer of the library's own: the GPU
// and its copy engines then never touch the caller's pages.  Left to the runtime, such a copy pins the caller's buffer in
// place for its duration; three runs of the GPU test suite in ~35 (rounds 4 and 5) died with "Memory access fault by GPU
// ... on address <a page boundary inside the process's heap>" while the main thread sat in exactly such a copy (a
// DeviceMatrix upload, kamd_nnet_forward, kamd_pipeline_load_batch), i.e. something reached one page past what had been
// pinned.  Page-locked and registered host memory (hipHostMalloc, hipHostRegister: the bench's uploads, the descriptor
// rings) and device-to-device copies pass through unchanged.  An "async" copy of pageable memory is synchronous here, as
// it effectively is in the runtime.  Every hipMemcpy / hipMemcpyAsync / hipMemcpy2D / hipMemcpy2DAsync of the library is one of
// these four (macros below), the exported kamd_memcpy_h2d / kamd_memcpy_d2h included.
hipError_t MemcpySafe(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t MemcpyAsyncSafe(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t Memcpy2DSafe(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t Memcpy2DAsyncSafe(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t st);

// The page-locked host memory the copies above may read or write in place is the library's own: every hipHostMalloc /
// hipHostRegister of the library goes through these (macros below) and is noted in a list; everything else is pageable to
// the wrappers (common.cc says why the runtime is not asked).
void NotePinned(const void *p, size_t bytes);
void ForgetPinned(const void *p);
bool IsNotedPinned(const void *p, size_t bytes);
hipError_t HostMallocNoted(void **p, size_t bytes, unsigned flags);
hipError_t HostFreeNoted(void *p);
hipError_t HostRegisterNoted(void *p, size_t bytes, unsigned flags);
hipError_t HostUnregisterNoted(void *p);
template <typename T>
inline hipError_t HostMallocNotedT(T **p, size_t bytes, unsigned flags = hipHostMallocDefault) {
  return HostMallocNoted(reinterpret_cast<void **>(p), bytes, flags);
}

template <typename T>
inline T *DevAlloc(size_t n) {
  void *p = NULL;
  if (n == 0) n = 1;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return NULL;
  return static_cast<T *>(p);
}

inline int CeilDiv(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }
inline int RoundUp(int a, int b) { return ((a + b - 1) / b) * b; }

// ---- device helpers -------------------------------------------------------
// order-preserving float <-> uint map (for atomicMin on costs, radix select)
__host__ __device__ inline uint32_t FloatToOrdered(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float OrderedToFloat(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace kamd

#ifndef KAMD_RAW_MEMCPY          // (common.cc itself calls the runtime's functions)
#define hipMemcpy(...) kamd::MemcpySafe(__VA_ARGS__)
#define hipMemcpyAsync(...) kamd::MemcpyAsyncSafe(__VA_ARGS__)
#define hipMemcpy2D(...) kamd::Memcpy2DSafe(__VA_ARGS__)
#define hipMemcpy2DAsync(...) kamd::Memcpy2DAsyncSafe(__VA_ARGS__)
#define hipHostMalloc(...) kamd::HostMallocNotedT(__VA_ARGS__)
#define hipHostFree(p) kamd::HostFreeNoted(p)
#define hipHostRegister(...) kamd::HostRegisterNoted(__VA_ARGS__)
#define hipHostUnregister(p) kamd::HostUnregisterNoted(p)
#endif
#endif
