// Per-launch descriptors (row offsets, item lengths, utterance records) on their way to the device.
//
// A hipMemcpyAsync of a few hundred KB queues on the copy engine behind whatever bulk upload is in flight (the test
// set's 1.2 GB of waveforms: 21 ms on PCIe 5 x16) and, from pageable memory, blocks the host until the stream has
// drained -- measured on the headline step as 21 ms of idle device in front of the first acoustic-model pass and as
// one pass issued only after the previous one had run.  Here the descriptors are written into page-locked host memory
// and PULLED into HBM by a small kernel on the launch's own stream: no copy engine, no host wait.  A ring of slots so
// that several launches can be issued before the first has run; a slot is reused once the event recorded behind its
// last reader has passed.
#ifndef KALDI_AMD_CSRC_META_RING_H_
#define KALDI_AMD_CSRC_META_RING_H_

#include <hip/hip_runtime.h>

#include <cstddef>

namespace kamd {

class MetaRing {
 public:
  static constexpr int kSlots = 4;
  MetaRing() {}
  ~MetaRing();
  MetaRing(const MetaRing &) = delete;
  MetaRing &operator=(const MetaRing &) = delete;
  // Copies `bytes` (rounded up to 8) from `src` to a device buffer, on stream `st`; kernels launched on `st` after this
  // call see the data at *d.  Returns a KAMD status.  Every Acquire is followed by one Release.
  int Acquire(const void *src, size_t bytes, void **d, hipStream_t st);
  // The same in two steps, for descriptors that hold pointers INTO the device buffer: Reserve hands out the slot's host
  // buffer (*h, to be filled by the caller) and the device address the bytes will have (*d); Commit sends them.
  int Reserve(size_t bytes, void **h, void **d);
  int Commit(hipStream_t st);
  // The kernels issued on `st` so far are the slot's last readers.
  int Release(hipStream_t st);

 private:
  void *h_[kSlots] = {NULL, NULL, NULL, NULL};
  void *d_[kSlots] = {NULL, NULL, NULL, NULL};
  void *h_dev_[kSlots] = {NULL, NULL, NULL, NULL};     // h_ as the device addresses it
  size_t cap_ = 0;                                     // 8-byte words, every slot
  hipEvent_t ev_[kSlots] = {NULL, NULL, NULL, NULL};
  bool busy_[kSlots] = {false, false, false, false};
  unsigned next_ = 0;
  int cur_ = -1;
  size_t cur_words_ = 0;
  bool committed_ = false;
};

}  // namespace kamd

#endif  // KALDI_AMD_CSRC_META_RING_H_
