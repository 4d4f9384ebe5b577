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
#endif
