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
// optimizers with a second slot: Adam (v), Adadelta (accum_update), FTRL (linear)
inline bool opt_needs_v(int kind) { return kind == 1 || kind == 4 || kind == 5; }

// Times everything enqueued on `stream` during its lifetime when xdfm_prof_enable(1) is on
// (two HIP events, owned by the scope until both are recorded); a no-op otherwise and on a stream
// that is being captured into a graph.  The name is the kernel's (as rocprofv3 prints it).
class ProfScope {
 public:
  ProfScope(const char* name, void* stream);
  ~ProfScope();
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;

 private:
  const char* name_;
  void* stream_;
  void* beg_;  // hipEvent_t, null when the scope is off
  void* end_;
};

}  // namespace xdfm

// launch `kernel` under a timing scope named after it
#define XDFM_LAUNCH(name, stream, kernel, grid, block, lds, ...)                            \
  do {                                                                                      \
    xdfm::ProfScope prof_scope_(name, (void*)(stream));                                     \
    hipLaunchKernelGGL(kernel, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__);       \
  } while (0)

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

// Wave priority of the small kernels on the step's side chains (DNN tower, statistics, loss): they share
// the chip with a matrix-bound CIN kernel whose waves issue back to back, and each of them gates the next
// node of its chain.  (A build-time knob for A/B runs; 0 = leave the default.)
#ifndef XDFM_SIDE_PRIO
#define XDFM_SIDE_PRIO 0
#endif
#if defined(__HIP__)
__device__ __forceinline__ void side_chain_priority() {
#if XDFM_SIDE_PRIO
  __builtin_amdgcn_s_setprio(XDFM_SIDE_PRIO);
#endif
}
#endif
