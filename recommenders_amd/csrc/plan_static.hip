// xdfm_embed_plan_static: the id sort of xdfm_embed_plan for callers that cannot allocate (code
// under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
// in the reference.  The sort's scratch (ping-pong pair buffers + digit table) lives in a
// file-scope __device__ array of a fixed maximum size instead of a caller workspace; every GPU the
// library runs on holds its own copy until the process ends.
//
// The array's NAME is a host-side handle, not a device address: the entry point resolves it once
// per device with hipGetSymbolAddress (that device current) and hands only that pointer to the
// kernels.  All calls on one GPU share the array: calls that overlap on one GPU (two streams, two
// threads) race.  Nothing in it has to start at zero: every pass writes the digit table and the
// pair buffers before it reads them.
#include <hip/hip_runtime.h>

#include <mutex>

#include "sort_plan.h"
#include "xdfm_common.h"

namespace {

constexpr int64_t kStaticMaxTokens = (int64_t)1 << 19;  // 524 288 tokens = 13 443 samples of 39 fields
constexpr size_t kStaticBytes = xdfm::plan_sort_workspace_bytes(kStaticMaxTokens);  // 9 MiB
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
std::mutex g_mu;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

}  // namespace

extern "C" int xdfm_embed_plan_static(const int64_t* ids, int64_t nnz, int64_t N, int32_t* sorted_ids,
                                      int32_t* sorted_tok, void* stream) {
  XDFM_REQUIRE(nnz >= 0 && N > 0 && N <= ((int64_t)1 << 31), "xdfm_embed_plan_static: bad sizes");
  if (nnz > kStaticMaxTokens)
    return xdfm::fail(XDFM_E_WORKSPACE,
                      "xdfm_embed_plan_static: %lld tokens exceed the static scratch's maximum of %lld; use "
                      "xdfm_embed_plan with a caller workspace",
                      (long long)nnz, (long long)kStaticMaxTokens);
  if (nnz == 0) return XDFM_OK;
  int dev = -1;
  XDFM_HIP(hipGetDevice(&dev));
  XDFM_REQUIRE(dev >= 0 && dev < kMaxDevices, "xdfm_embed_plan_static: device index %d out of range", dev);
  void* ptr = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_scratch_ptr[dev]) {
      void* p = nullptr;
      XDFM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      XDFM_REQUIRE(p != nullptr, "xdfm_embed_plan_static: no device address for the scratch array");
      g_scratch_ptr[dev] = p;
    }
    ptr = g_scratch_ptr[dev];
  }
  return xdfm::plan_sort(ids, nnz, N, sorted_ids, sorted_tok, ptr, kStaticBytes, (hipStream_t)stream);
}
