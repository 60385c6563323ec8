// Host-side error plumbing + misc C-ABI entry points.
#include "common.hpp"

#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

namespace kvq {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return KVQ_ERR_HIP;
}
// Keyed by (kernel address, device ordinal), not by the call site: a generic lambda `go(auto k)` over kernels that all decay to the same
// pointer type is instantiated ONCE, so a `static LdsOptIn` inside it is shared by every kernel passed through it (embed.hip, conv.hip,
// slowneck.hip) — the first kernel's opt-in must not stand for the others.
int LdsOptIn::ensure(const void* kernel, int want) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> granted;
  int dev = 0;
  KVQ_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int& have = granted[std::make_pair(kernel, dev)];
  if (have >= want) return KVQ_OK;
  KVQ_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want));
  have = want;
  return KVQ_OK;
}
bool latency_mode() {
  static const bool on = getenv("KVQ_LATENCY") && atoi(getenv("KVQ_LATENCY")) == 1;
  return on;
}
}  // namespace kvq

extern "C" int kvq_abi_version(void) { return KVQ_ABI_VERSION; }
extern "C" const char* kvq_last_error(void) { return kvq::g_err; }

extern "C" int kvq_device_name(char* buf, int n) {
  if (!buf || n <= 0) return KVQ_ERR_NULL;
  buf[0] = 0;
  int count = 0;
  KVQ_CHECK_HIP(hipGetDeviceCount(&count));
  KVQ_REQUIRE(count > 0, KVQ_ERR_HIP, "kvq_device_name: no HIP device visible");
  int dev = 0;
  KVQ_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  KVQ_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return KVQ_OK;
}
