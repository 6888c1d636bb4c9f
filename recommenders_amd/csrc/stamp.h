// Measurement only: device-side time stamps of the step's kernels (xdfm_debug_stamps, include/xdfm.h).
// A replayed HIP graph cannot be seen from outside without changing it: the tracer's per-dispatch signals
// make the launch host-bound and HIP events cannot be read back from inside a capture.  So a kernel that
// carries `Stamp _s(kStampX);` as its first statement writes the constant 100 MHz clock (s_memrealtime) into
// buf[2 id] (first block, minimum over launches = start) and buf[2 id + 1] (maximum over a sample of the
// blocks that includes the last ones dispatched = end) when a buffer is installed, and costs one scalar load and a branch at either end when none is.
//
// Each translation unit holds its own copy of the pointer (no relocatable device code in this build):
// XDFM_STAMP_UNIT(name) defines the setter of the unit, runtime.cpp calls all of them.
#pragma once
#include <hip/hip_runtime.h>

namespace xdfm {

enum StampId {
  kStampEmbedFwd = 0,
  kStampCinFwd,
  kStampPooledFwd,
  kStampPooledBwd,
  kStampCinDx,
  kStampCinDw,
  kStampEmbedBwd,
  kStampEmbedBwdLong,
  kStampDenseUpdate,
  kStampDenseFwd,
  kStampDenseBwdDx,
  kStampDenseBwdDw,
  kStampSort,
  kStampPlanGather,
  kStampCinPack,
  kStampLossFinish,
  kStampOptAdvance,
  kStampDenseStats,
  kStampCinDwUnpack,
  kStampLossPair,
  kStampPooledDwReduce,
  kStampRoute,          // route_keys / count / write / pad / decode
  kStampGatherRows,     // the owner-side row gather of the sharded table
  kStampEmbedScatter,   // embed_bwd in a scatter mode (per-slot gradient records, dense gradient image)
  kStampCount
};

#if defined(__HIP__)
namespace {
__device__ unsigned long long* g_stamp_buf = nullptr;

// One atomic per launch for the start (block 0) and a few dozen for the end (every 64th block and the last 8
// in dispatch order): atomics of every block to one address queue up behind each other, ~0.1 us apiece, and
// would stretch the kernels they are meant to time.
struct Stamp {
  unsigned long long* p;
  __device__ __forceinline__ explicit Stamp(int id) {
    unsigned long long* b = g_stamp_buf;
    p = nullptr;
    if (b && threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
      const unsigned n = gridDim.x * gridDim.y * gridDim.z;
      const unsigned i = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      if (i == 0) atomicMin(b + 2 * id, (unsigned long long)__builtin_amdgcn_s_memrealtime());
      if ((i & 63) == 0 || i + 8 >= n) p = b + 2 * id;
    }
  }
  __device__ __forceinline__ ~Stamp() {
    if (p) atomicMax(p + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
};
}  // namespace

#define XDFM_STAMP_UNIT(name)                                                                          \
  namespace xdfm {                                                                                     \
  int name(unsigned long long* buf) {                                                                  \
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf), 0, hipMemcpyHostToDevice); \
  }                                                                                                    \
  }
#endif

}  // namespace xdfm
