// Library-level entry points: ABI version and the last-error string (thread local).
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

void vr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int vrnet_abi_version(void) { return 3; }
extern "C" const char* vrnet_last_error(void) { return g_err; }

// Synchronous device query used by load-time checks only (never on the hot path).
extern "C" int vrnet_device_arch(char* buf, int len) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    vr_set_error("no HIP device");
    return VR_ERR_LAUNCH;
  }
  strncpy(buf, prop.gcnArchName, len - 1);
  buf[len - 1] = 0;
  return VR_OK;
}
