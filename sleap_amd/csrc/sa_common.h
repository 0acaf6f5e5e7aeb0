// Shared host-side plumbing for libsleap_amd.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/sleap_amd.h"

namespace sa {

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace sa

// internal (not ABI): the workgroup limit of sa_conv3x3_set_grid_limit (conv3x3.hip), honoured by every persistent kernel
int sa_internal_grid_limit();

#define SA_HIP_CHECK(expr)                                                                   \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return sa::fail(SA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                      __FILE__, __LINE__);                                                   \
  } while (0)

#define SA_LAUNCH_CHECK() SA_HIP_CHECK(hipGetLastError())

#define SA_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return sa::fail(SA_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)
