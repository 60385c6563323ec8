// Host-side error plumbing + misc C-ABI entry points.
#include "common.hpp"

#include <string.h>

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
int LdsOptIn::ensure(const void* kernel, int want) {
  int dev = 0;
  KVQ_CHECK_HIP(hipGetDevice(&dev));
  const bool known = dev >= 0 && dev < 16;
  if (known && bytes[dev] >= want) return KVQ_OK;
  KVQ_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want));
  if (known) bytes[dev] = want;
  return KVQ_OK;
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
