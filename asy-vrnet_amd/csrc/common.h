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
// Tuning knobs and timing ablations exist ONLY in the diagnostic build (make tuning -> libvrnet_hip_tuning.so, compiled
// with -DVR_TUNING and loaded through VRNET_HIP_LIB): the product library reads no environment variable at all.
//   vr_tune("VRNET_X", d): integer knob, d in the product build.
//   vr_ablated(group): VRNET_ABLATE = comma list of kernel groups whose launches are skipped (results are garbage,
//   timing valid) -- igemm, wgrad, moments, affine, cluster, coef, spatial; always false in the product build.
#ifdef VR_TUNING
int vr_tune(const char* name, int dflt);
bool vr_ablated(const char* group);
#else
static inline int vr_tune(const char*, int dflt) { return dflt; }
static inline bool vr_ablated(const char*) { return false; }
#endif

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

extern "C" {
/* include/vrnet_hip.h: optional bf16-plane output of a producing kernel (csrc/pgemm.hip: plane tensors). */
typedef struct vrnet_planes_out {
  void* p;          /* plane q of element (row, c) at p[q * plane + row * ld + c], bf16 */
  long ld, plane;   /* row and plane stride in elements (multiples of 4; 8 to feed the plane GEMMs) */
  int np;           /* 3: the fp32 value split exactly into three planes; 1: rounded to bf16 */
} vrnet_planes_out;
}
static inline bool vr_planes_out_ok(const vrnet_planes_out* o, int C) {
  return !o || (o->p && (o->np == 1 || o->np == 3) && o->ld >= C && o->ld % 4 == 0 && (o->np == 1 || o->plane % 4 == 0) &&
                (reinterpret_cast<uintptr_t>(o->p) & 7) == 0);
}

// Four consecutive fp32 values -> bf16 planes (8 bytes per plane): np = 3 splits each value exactly, v = p0 + p1 + p2 with
// round-to-nearest-even at each step (the residuals v - p0 and v - p0 - p1 are exact in fp32 and the last one has at most 8
// significant bits); np = 1 rounds to bf16.  All arithmetic per component (no packed fp32 with operand broadcast).
typedef __bf16 vr_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vr_store_planes4(unsigned short* dst, long plane, int np, const f32x4 v) {
  const vr_bf16x4 p0 = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<vr_bf16x4*>(dst) = p0;
  if (np == 3) {
    const float r0 = v[0] - (float)p0[0], r1 = v[1] - (float)p0[1], r2 = v[2] - (float)p0[2], r3 = v[3] - (float)p0[3];
    const vr_bf16x4 p1 = {(__bf16)r0, (__bf16)r1, (__bf16)r2, (__bf16)r3};
    *reinterpret_cast<vr_bf16x4*>(dst + plane) = p1;
    const vr_bf16x4 p2 = {(__bf16)(r0 - (float)p1[0]), (__bf16)(r1 - (float)p1[1]), (__bf16)(r2 - (float)p1[2]),
                          (__bf16)(r3 - (float)p1[3])};
    *reinterpret_cast<vr_bf16x4*>(dst + 2 * plane) = p2;
  }
}


__device__ __forceinline__ float vr_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
// exact-erf GELU (nn.GELU default, vr_coc.py:204) and its derivative.  erf is evaluated branch-free as
//   erf(z) = 1 - Q(s) exp(-z^2),  s = p z / (1 + p z),  Q a degree-6 polynomial with Q(0) = 1      (z = |u| / sqrt 2)
// (the Abramowitz-Stegun 7.1.26 form, re-fitted: 1.5e-9 maximum error in exact arithmetic, tools/fit_erf.py).  In fp32 the
// resulting GELU / GELU' deviate from the exact functions by at most 3.0e-7 / 1.4e-7 absolute -- the same as torch's own
// 0.5 x (1 + erff(x / sqrt 2)) in fp32 (4.5e-7 / 1.4e-7) -- at 18 VALU operations instead of the ~55 of the library erff with
// its two divergent branches (the GELU epilogues were VALU-bound on it).  Written in s rather than t = 1 / (1 + p z): near
// z = 0 a rounding error in t is amplified by Q'(0) ~ 3.4, an error in s is proportional to z.  exp(-z^2) = exp(-u^2 / 2) is
// also the Gaussian of the derivative.
__device__ __forceinline__ float vr_erf_abs(float u, float& gauss) {      // erf(|u| / sqrt 2); gauss = exp(-u^2 / 2)
  // (clamped: u = +-Inf must give s = 1, not Inf * 0; a NaN still propagates through the Gaussian)
  const float pz = fminf(fabsf(u) * (0.37458086389741496f * 0.70710678118654752f), 1e30f);
  const float s = pz * __builtin_amdgcn_rcpf(pz + 1.0f);
  gauss = __builtin_amdgcn_exp2f((u * u) * -0.72134752044448170f);
  float q = -0.30087511512911436f;
  q = fmaf(q, s, 0.4339584527532372f);
  q = fmaf(q, s, 0.8220608039774832f);
  q = fmaf(q, s, -3.070689406167805f);
  q = fmaf(q, s, 4.114633908437599f);
  q = fmaf(q, s, -3.012377713453634f);
  q = fmaf(q, s, 1.0f);
  return fmaf(-q, gauss, 1.0f);
}
__device__ __forceinline__ float vr_gelu(float x) {        // 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + 0.5 |x| erf(|x| / sqrt 2)
  float g;
  const float e = vr_erf_abs(x, g);
  return fmaf(0.5f * fabsf(x), e, 0.5f * x);
}
// gelu(x) and gelu'(x) = cdf + x pdf from one erf evaluation
__device__ __forceinline__ float vr_gelu_both(float x, float& grad) {
  float g;
  const float e = vr_erf_abs(x, g);
  const float cdf = 0.5f + copysignf(0.5f * e, x);
  grad = fmaf(x, 0.39894228040143268f * g, cdf);
  return x * cdf;
}
__device__ __forceinline__ float vr_gelu_grad(float x) {
  float grad;
  vr_gelu_both(x, grad);
  return grad;
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
