// kamd::MetaRing: see meta_ring.h.
#include "meta_ring.h"

#include <algorithm>
#include <cstdint>
#include <cstring>

#include "common.h"

namespace kamd {

__global__ __launch_bounds__(256) void MetaPullKernel(uint64_t *dst, const uint64_t *src_host, size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) dst[i] = src_host[i];
}

MetaRing::~MetaRing() {
  for (int r = 0; r < kSlots; r++) {
    if (ev_[r]) {
      if (busy_[r]) (void)hipEventSynchronize(ev_[r]);
      (void)hipEventDestroy(ev_[r]);
    }
    if (d_[r]) (void)hipFree(d_[r]);
    if (h_[r]) (void)hipHostFree(h_[r]);
  }
}

int MetaRing::Reserve(size_t bytes, void **h, void **d) {
  if (cur_ >= 0) return SetError(KAMD_ERR_STATE, "MetaRing: Reserve without Release");
  const int r = static_cast<int>(next_++ % kSlots);
  const size_t words = std::max<size_t>(1, (bytes + 7) / 8);
  if (words > cap_) {
    // all slots at once, so that launches of alternating sizes (the passes of a test set) stop allocating after the first
    // round: page-locking and hipFree take milliseconds and wait for the device
    for (int q = 0; q < kSlots; q++) {
      if (busy_[q]) { KAMD_HIP(hipEventSynchronize(ev_[q])); busy_[q] = false; }
      if (!ev_[q]) KAMD_HIP(hipEventCreateWithFlags(&ev_[q], hipEventDisableTiming));
      if (d_[q]) (void)hipFree(d_[q]);
      if (h_[q]) (void)hipHostFree(h_[q]);
      d_[q] = NULL; h_[q] = NULL;
    }
    cap_ = 0;
    const size_t cap = std::max<size_t>(words + words / 4, 1024);
    for (int q = 0; q < kSlots; q++) {
      KAMD_HIP(hipHostMalloc(&h_[q], cap * 8, hipHostMallocMapped));
      KAMD_HIP(hipMalloc(&d_[q], cap * 8));
      KAMD_HIP(hipHostGetDevicePointer(&h_dev_[q], h_[q], 0));
    }
    cap_ = cap;
  }
  if (busy_[r]) { KAMD_HIP(hipEventSynchronize(ev_[r])); busy_[r] = false; }      // the launch that used this slot has run
  static_cast<uint64_t *>(h_[r])[words - 1] = 0;
  cur_ = r; cur_words_ = words; committed_ = false;
  *h = h_[r]; *d = d_[r];
  return KAMD_OK;
}

int MetaRing::Commit(hipStream_t st) {
  if (cur_ < 0 || committed_) return SetError(KAMD_ERR_STATE, "MetaRing: Commit without Reserve");
  const int r = cur_;
  hipLaunchKernelGGL(MetaPullKernel, dim3(std::min(256, CeilDiv(static_cast<int64_t>(cur_words_), 256))), dim3(256), 0, st,
                     static_cast<uint64_t *>(d_[r]), static_cast<const uint64_t *>(h_dev_[r]), cur_words_);
  KAMD_HIP(hipGetLastError());
  committed_ = true;
  return KAMD_OK;
}

int MetaRing::Acquire(const void *src, size_t bytes, void **d, hipStream_t st) {
  void *h = NULL;
  const int rc = Reserve(bytes, &h, d);
  if (rc != KAMD_OK) return rc;
  memcpy(h, src, bytes);
  const int crc = Commit(st);
  if (crc != KAMD_OK) cur_ = -1;        // (nothing was sent: the slot is free again, the caller has no Release to make)
  return crc;
}

int MetaRing::Release(hipStream_t st) {
  if (cur_ < 0) return KAMD_OK;
  const int r = cur_;
  cur_ = -1;
  if (!committed_) return KAMD_OK;            // reserved, never sent: nothing on the device reads the slot
  KAMD_HIP(hipEventRecord(ev_[r], st));
  busy_[r] = true;
  return KAMD_OK;
}

}  // namespace kamd
