// Host runtime pieces of libxdfm.so shared by every entry point: the per-thread error
// message and the optional per-kernel timing (HIP events on the launch stream) that
// bench.py uses for its roofline line.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "xdfm_common.h"

namespace xdfm {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct Record {
  const char* name;
  hipEvent_t beg, end;
};
std::mutex g_mu;
std::atomic<bool> g_on{false};
std::vector<Record> g_records;
constexpr size_t kMaxRecords = 1u << 18;
}  // namespace

ProfScope::ProfScope(const char* name, void* stream) : name_(name), stream_(stream), beg_(nullptr), end_(nullptr) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  // an event recorded on a capturing stream would become part of the graph (or fail the capture)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;
  }
  hipEvent_t beg = nullptr, end = nullptr;
  if (hipEventCreate(&beg) != hipSuccess) return;
  if (hipEventCreate(&end) != hipSuccess) {
    (void)hipEventDestroy(beg);
    return;
  }
  (void)hipEventRecord(beg, (hipStream_t)stream);
  beg_ = beg;
  end_ = end;
}

ProfScope::~ProfScope() {
  if (!beg_) return;
  (void)hipEventRecord((hipEvent_t)end_, (hipStream_t)stream_);
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_on.load(std::memory_order_relaxed) && g_records.size() < kMaxRecords) {
    g_records.push_back(Record{name_, (hipEvent_t)beg_, (hipEvent_t)end_});  // complete: both events recorded
  } else {
    (void)hipEventDestroy((hipEvent_t)beg_);
    (void)hipEventDestroy((hipEvent_t)end_);
  }
}

}  // namespace xdfm

namespace xdfm {
int stamp_unit_embed(unsigned long long*);
int stamp_unit_cin(unsigned long long*);
int stamp_unit_pooled(unsigned long long*);
int stamp_unit_dnn(unsigned long long*);
int stamp_unit_optim(unsigned long long*);
int stamp_unit_sort(unsigned long long*);
int stamp_unit_route(unsigned long long*);
}  // namespace xdfm

extern "C" int xdfm_debug_stamps(uint64_t* device_buf) {
  unsigned long long* b = reinterpret_cast<unsigned long long*>(device_buf);
  int (*const units[])(unsigned long long*) = {xdfm::stamp_unit_embed, xdfm::stamp_unit_cin,  xdfm::stamp_unit_pooled,
                                               xdfm::stamp_unit_dnn,   xdfm::stamp_unit_optim, xdfm::stamp_unit_sort,
                                               xdfm::stamp_unit_route};
  for (auto u : units) {
    const int e = u(b);
    if (e) return xdfm::fail(XDFM_E_HIP, "xdfm_debug_stamps: %s", hipGetErrorString((hipError_t)e));
  }
  return XDFM_OK;
}

extern "C" int xdfm_abi_version(void) { return XDFM_ABI_VERSION; }
extern "C" const char* xdfm_last_error(void) { return xdfm::error_buffer(); }

extern "C" int xdfm_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(xdfm::g_mu);
  xdfm::g_on.store(on != 0);
  return XDFM_OK;
}

extern "C" int xdfm_prof_collect(char* json, size_t cap) {
  XDFM_REQUIRE(json && cap >= 64, "xdfm_prof_collect: buffer too small");
  std::vector<xdfm::Record> recs;
  {
    std::lock_guard<std::mutex> lk(xdfm::g_mu);
    recs.swap(xdfm::g_records);
  }
  struct Agg {
    double ms = 0;
    long n = 0;
  };
  std::map<std::string, Agg> agg;
  int rc = XDFM_OK;
  for (auto& r : recs) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.end);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.beg, r.end);
    if (e == hipSuccess) {
      Agg& a = agg[r.name];
      a.ms += ms;
      a.n += 1;
    } else {
      rc = xdfm::fail(XDFM_E_HIP, "xdfm_prof_collect: %s", hipGetErrorString(e));
    }
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  std::string out = "{";
  bool first = true;
  for (auto& kv : agg) {
    char line[256];
    snprintf(line, sizeof(line), "%s\"%s\": {\"ms\": %.6f, \"n\": %ld}", first ? "" : ", ",
             kv.first.c_str(), kv.second.ms, kv.second.n);
    out += line;
    first = false;
  }
  out += "}";
  if (out.size() + 1 > cap) return xdfm::fail(XDFM_E_WORKSPACE, "xdfm_prof_collect: need %zu bytes", out.size() + 1);
  memcpy(json, out.c_str(), out.size() + 1);
  return rc;
}
