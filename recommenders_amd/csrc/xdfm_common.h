// Shared host-side helpers of libxdfm.so (error reporting, launch checks).
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/xdfm.h"

namespace xdfm {

// thread-local message returned by xdfm_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

}  // namespace xdfm

#define XDFM_REQUIRE(cond, ...)                              \
  do {                                                       \
    if (!(cond)) return xdfm::fail(XDFM_E_INVALID, __VA_ARGS__); \
  } while (0)

#define XDFM_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess)                                                           \
      return xdfm::fail(XDFM_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                        \
  } while (0)

#define XDFM_LAUNCH_CHECK() XDFM_HIP(hipGetLastError())
