#ifndef FAKE_COMMON_H_
#define FAKE_COMMON_H_
#include <stdint.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include "../../include/kaldi_amd.h"
namespace kamd { int SetError(int code, const char *fmt, ...); }
#endif
