// Shared helpers for libvrnet_hip.so (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define VR_OK 0
#define VR_ERR_ARG 1
#define VR_ERR_LAUNCH 2
#define VR_ERR_WORKSPACE 3

void vr_set_error(const char* fmt, ...);
void vr_note_kernel(int id);
// Timing ablation (diagnostic only, results are garbage): VRNET_ABLATE = comma list of kernel groups whose launches are
// skipped -- igemm, wgrad, moments, affine, cluster, coef, spatial.  Tells how much of the step each group exposes.
bool vr_ablated(const char* group);

#define VR_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      vr_set_error(__VA_ARGS__);                \
      return VR_ERR_ARG;                        \
    }                                           \
  } while (0)

#define VR_LAUNCH_CHECK(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      vr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return VR_ERR_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

static inline hipStream_t vr_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long vr_cdiv(long a, long b) { return (a + b - 1) / b; }
static inline bool vr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float vr_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// exact-erf GELU (nn.GELU default) and its derivative
__device__ __forceinline__ float vr_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float vr_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
