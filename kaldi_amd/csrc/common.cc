#include "common.h"

namespace kamd {
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
int kamd_memcpy_h2d(void *d, const void *h, size_t bytes) { KAMD_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return KAMD_OK; }
int kamd_memcpy_d2h(void *h, const void *d, size_t bytes) { KAMD_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return KAMD_OK; }
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
}
