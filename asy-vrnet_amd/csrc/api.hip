// Library-level entry points: ABI version and the last-error string (thread local).
#include "common.h"

#include <atomic>
#include <cstring>

static thread_local char g_err[512] = "";
static thread_local int g_last_kernel = 0;
static std::atomic<long> g_kernel_count[16];      // process-wide: the backward pass runs on autograd's own thread
void vr_note_kernel(int id) {
  g_last_kernel = id;
  if (id >= 0 && id < 16) ++g_kernel_count[id];
}
#ifdef VR_TUNING
#include <cstdlib>
int vr_tune(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
bool vr_ablated(const char* group) {
  static const char* env = getenv("VRNET_ABLATE");
  if (env == nullptr) return false;
  const size_t n = strlen(group);
  for (const char* t = env; (t = strstr(t, group)) != nullptr; t += n)      // whole comma-separated tokens only
    if ((t == env || t[-1] == ',') && (t[n] == 0 || t[n] == ',' || t[n] == ' ')) return true;
  return false;
}
/* 1 in the diagnostic build (environment knobs and VRNET_ABLATE honoured), 0 in the product library. */
extern "C" int vrnet_tuning_build(void) { return 1; }
#else
extern "C" int vrnet_tuning_build(void) { return 0; }
#endif
// Which kernel family the last vrnet_conv2d_f32 / vrnet_conv2d_wgrad_f32 call of this thread dispatched to (bench.py
// prices each launch against the roofline of the kernel that actually ran):
//   1 fp32 MFMA register-staged   2 fp32 MFMA LDS-DMA ring   3 bf16-rounded operands   4 direct (tiny channel counts)
//   6 six exact bf16 x bf16 products per fp32 product (x6), LDS-DMA ring
extern "C" int vrnet_last_kernel(void) { return g_last_kernel; }
/* Launches of kernel family `family` (the codes of vrnet_last_kernel) issued by this process since the library was loaded
 * (all threads: autograd runs the backward pass on a thread of its own): lets a caller assert which kernels a whole
 * forward / backward pass actually ran on. */
extern "C" long vrnet_kernel_launches(int family) { return (family >= 0 && family < 16) ? g_kernel_count[family].load() : 0; }

void vr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Diagnostic: one thread writes the device's constant-rate clock (100 MHz) to *dst at the point of `stream` where it is issued --
// timelines of the captured step without a tracer attached (tools/debug/section_stamps.py).
__global__ void clock_stamp_kernel(long long* dst) { *dst = (long long)wall_clock64(); }
extern "C" int vrnet_clock_stamp(long long* dst, void* stream) {
  if (!dst) return VR_ERR_ARG;
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  return VR_OK;
}

extern "C" int vrnet_abi_version(void) { return 10; }
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
