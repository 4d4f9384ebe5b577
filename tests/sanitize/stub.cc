// host-only build of the I/O + determinization sources for sanitizer runs
#include "../../include/kaldi_amd.h"
#include <cstdarg>
#include <cstdio>
namespace kamd {
static thread_local char g_err[1024];
int SetError(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); return code; }
}
extern "C" const char *kamd_last_error(void) { return kamd::g_err; }
extern "C" kamd_graph *kamd_graph_create(int32_t, int32_t, const int64_t *, const kamd_arc *, const float *) { return 0; }
