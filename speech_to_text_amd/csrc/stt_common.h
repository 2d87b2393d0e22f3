// Shared host-side helpers of libstt_mi355x.so: error reporting.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/stt_mi355x.h"

// Records a message for stt_last_error() and returns `code`.
int stt_fail(int code, const char* fmt, ...);

#define STT_HIP_OR_FAIL(expr, cleanup)                                                  \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      cleanup;                                                                          \
      return stt_fail(STT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                              \
    }                                                                                   \
  } while (0)
