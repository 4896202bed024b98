// Library-level entry points: ABI version and the last-error string (thread local).
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";
static thread_local int g_last_kernel = 0;
void vr_note_kernel(int id) { g_last_kernel = id; }
bool vr_ablated(const char* group) {
  static const char* env = getenv("VRNET_ABLATE");
  if (env == nullptr) return false;
  const size_t n = strlen(group);
  for (const char* t = env; (t = strstr(t, group)) != nullptr; t += n)      // whole comma-separated tokens only
    if ((t == env || t[-1] == ',') && (t[n] == 0 || t[n] == ',' || t[n] == ' ')) return true;
  return false;
}
// Which kernel family the last vrnet_conv2d_f32 / vrnet_conv2d_wgrad_f32 call of this thread dispatched to (bench.py
// prices each launch against the roofline of the kernel that actually ran):
//   1 fp32 MFMA register-staged   2 fp32 MFMA LDS-DMA ring   3 bf16-rounded operands   4 direct (tiny channel counts)
//   6 six exact bf16 x bf16 products per fp32 product (x6), LDS-DMA ring
extern "C" int vrnet_last_kernel(void) { return g_last_kernel; }

void vr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int vrnet_abi_version(void) { return 4; }
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
