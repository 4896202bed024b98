// Fused streaming kernels of the asymmetric fusion blocks (round 5): ImageEnhanceByRadar / RadarEnhanceByImage
// (backbone/fusion/vr_coc.py:303-359) are chains of small elementwise passes and per-channel reductions, and since the
// radar stream became the step's critical chain every launch of them is exposed.  Each kernel here is an elementwise pass
// that ALSO leaves the partial sums its successor's reduction needs, so that the successor's own pass over the tensor (a
// launch + a read of everything just written) disappears:
//   bn_relu_minmax     p = ReLU(BN(z))                         + per-workgroup (min, max) of p      (data_normal, :59-67)
//   enhance_stats      t = (1 + data_normal(p)) * x            + column (sum, sumsq) of t           (BatchNorm `norm`, :315)
//   bn_relu_res_stats  s = ReLU(BN(z)) + r                     + column (sum, sumsq) of s           (BatchNorm `norm`, :357)
//   bn_bwd_enhance     dt = BN-backward apply                  + the four sums of the gain's backward
//   enhance_bwd_stats  dx, dp of the gain                      + column (sum dp', sum dp' z) of the masked dp (BatchNorm bn1 backward)
// All of them: contiguous NHWC tensors (row stride == C), C % 4 == 0, C <= 1024, 16-byte aligned; a thread walks float4 elements
// e0, e0 + G, ... with G = (workgroups x 256) a multiple of C / 4, so its four channels never change: column sums accumulate in
// fp64 registers and meet through LDS in a fixed thread order (deterministic, no atomics).  Column partials have the layout
// [workgroup][C][2] that vrnet_bn_coef_{fwd,bwd}_from_chunks read.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float bn_pre(float A, float x, float S, float D) { return __builtin_fmaf(A, x - S, D); }

// per-thread column sums (four channels 4 q .. 4 q + 3, q = e0 mod CV) -> partial[blockIdx.x][C][2]
__device__ __forceinline__ void col_stats_flush(const double (&s1)[4], const double (&s2)[4], int CV, double* partial, double* lds) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lds[tid * 8 + j] = s1[j];
    lds[tid * 8 + 4 + j] = s2[j];
  }
  __syncthreads();
  if (tid < CV) {
    const int off = (int)(((long)blockIdx.x * 256) % CV);
    int t0 = tid - off;
    if (t0 < 0) t0 += CV;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    for (int t = t0; t < 256; t += CV)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] += lds[t * 8 + j];
        b[j] += lds[t * 8 + 4 + j];
      }
    double* o = partial + ((long)blockIdx.x * CV + tid) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[2 * j] = a[j];
      o[2 * j + 1] = b[j];
    }
  }
}

// every workgroup folds the (min, max) partial pairs (fminf / fmaxf: the same bits whatever the order)
__device__ __forceinline__ void minmax_fold(const float* partial, int nblocks, float* smn, float* smx, float& mn, float& mx) {
  mn = INFINITY;
  mx = -INFINITY;
  for (int e = threadIdx.x; e < nblocks; e += 256) {
    mn = fminf(mn, partial[2 * e]);
    mx = fmaxf(mx, partial[2 * e + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
}

// p = ReLU(A (z - S) + D); per-workgroup (min, max) of p.  res != nullptr: s = p + res with column (sum, sumsq) of s instead.
template <bool RES>
__global__ __launch_bounds__(256) void bn_relu_kernel(const float* __restrict__ z, const float* __restrict__ A, const float* __restrict__ D,
                                                      const float* __restrict__ S, const float* __restrict__ res, float* __restrict__ out,
                                                      long n4, int C, float* __restrict__ mmpart, double* __restrict__ colpart) {
  __shared__ double lds[RES ? 256 * 8 : 1];
  __shared__ float smn[4], smx[4];
  const int CV = C >> 2;
  const long G = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(e0 % CV) * 4;
  const f32x4 a4 = ld4(A + c0), d4 = ld4(D + c0), s4 = ld4(S + c0);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  float mn = INFINITY, mx = -INFINITY;
  for (long e = e0; e < n4; e += G) {
    const f32x4 zv = ld4(z + 4 * e);
    f32x4 r4 = {0.f, 0.f, 0.f, 0.f}, o;
    if (RES) r4 = ld4(res + 4 * e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = fmaxf(bn_pre(a4[j], zv[j], s4[j], d4[j]), 0.f);
      if (RES) {
        v += r4[j];
        s1[j] += (double)v;
        s2[j] += (double)v * (double)v;
      } else {
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
      }
      o[j] = v;
    }
    *reinterpret_cast<f32x4*>(out + 4 * e) = o;
  }
  if (RES) {
    col_stats_flush(s1, s2, CV, colpart, lds);
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, o, 64));
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      smn[threadIdx.x >> 6] = mn;
      smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      mmpart[2 * blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
      mmpart[2 * blockIdx.x + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
  }
}

// t = (1 + (p - mn) / (mx - mn)) * x with (mn, mx) folded from the partial pairs; column (sum, sumsq) of t
__global__ __launch_bounds__(256) void enhance_stats_kernel(const float* p, const float* x, const float* mmpart, int nmm, float* mm,
                                                            float* t, long n4, int C, double* colpart) {
  __shared__ double lds[256 * 8];
  __shared__ float smn[4], smx[4];
  float mn, mx;
  minmax_fold(mmpart, nmm, smn, smx, mn, mx);
  if (blockIdx.x == 0 && threadIdx.x == 0) { mm[0] = mn; mm[1] = mx; }
  const float dst = mx - mn;
  const int CV = C >> 2;
  const long G = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (long e = e0; e < n4; e += G) {
    const f32x4 pv = ld4(p + 4 * e), xv = ld4(x + 4 * e);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (1.f + (pv[j] - mn) / dst) * xv[j];
      s1[j] += (double)o[j];
      s2[j] += (double)o[j] * (double)o[j];
    }
    *reinterpret_cast<f32x4*>(t + 4 * e) = o;
  }
  col_stats_flush(s1, s2, CV, colpart, lds);
}

// dt = A g + E (t - S) + D (training-mode BatchNorm backward apply, per channel) and the four sums of the gain's backward over
// dn = dt * x: [0] sum dn, [1] sum dn (p - mn), [2] #(p == mn), [3] #(p == mx)  ->  sums4[workgroup][4]
__global__ __launch_bounds__(256) void bn_bwd_enhance_kernel(const float* g, const float* t, const float* A, const float* E,
                                                             const float* D, const float* S, const float* x, const float* p,
                                                             const float* mm, float* dt, long n4, int C, double* sums4) {
  __shared__ double red[4][4];
  const int CV = C >> 2;
  const long G = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(e0 % CV) * 4;
  const f32x4 a4 = ld4(A + c0), e4 = ld4(E + c0), d4 = ld4(D + c0), s4 = ld4(S + c0);
  const float mn = mm[0], mx = mm[1];
  double s[4] = {0, 0, 0, 0};
  for (long e = e0; e < n4; e += G) {
    const f32x4 gv = ld4(g + 4 * e), tv = ld4(t + 4 * e), xv = ld4(x + 4 * e), pv = ld4(p + 4 * e);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = a4[j] * gv[j];
      v += e4[j] * (tv[j] - s4[j]);
      v += d4[j];
      o[j] = v;
      const double dn = (double)v * (double)xv[j];
      s[0] += dn;
      s[1] += dn * (double)(pv[j] - mn);
      s[2] += (pv[j] == mn) ? 1.0 : 0.0;
      s[3] += (pv[j] == mx) ? 1.0 : 0.0;
    }
    *reinterpret_cast<f32x4*>(dt + 4 * e) = o;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[threadIdx.x >> 6][i] = s[i];
  __syncthreads();
  if (threadIdx.x < 4)
    sums4[4 * (long)blockIdx.x + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dx (+)= dt (1 + n(p)); dp = dt x / dst + [p == mn] g_mn + [p == mx] g_mx (the four sums folded from sums4 by every workgroup in
// the same order) and the column sums BatchNorm bn1's backward needs of dp' = dp [fA (z - fS) + fD > 0]: (sum dp', sum dp' z)
__global__ __launch_bounds__(256) void enhance_bwd_stats_kernel(const float* dt, const float* x, const float* p, const float* mm,
                                                                const double* sums4, int nsums, const float* z, const float* fA,
                                                                const float* fD, const float* fS, float* dx, float* dp, long n4,
                                                                int C, int accumulate_dx, double* colpart) {
  __shared__ double lds[256 * 8];
  __shared__ double red[4][4];
  __shared__ double sums[4];
  {
    double s[4] = {0, 0, 0, 0};
    for (int e = threadIdx.x; e < nsums; e += 256)
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += sums4[4 * (long)e + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[threadIdx.x >> 6][i] = s[i];
    __syncthreads();
    if (threadIdx.x < 4) sums[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
  }
  const float mn = mm[0], mx = mm[1];
  const double dstd = (double)mx - (double)mn;
  const float gmn = (float)((-sums[0] / dstd + sums[1] / (dstd * dstd)) / sums[2]);
  const float gmx = (float)((-sums[1] / (dstd * dstd)) / sums[3]);
  const float fd = (float)dstd;
  const int CV = C >> 2;
  const long G = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(e0 % CV) * 4;
  const f32x4 a4 = ld4(fA + c0), d4 = ld4(fD + c0), s4 = ld4(fS + c0);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (long e = e0; e < n4; e += G) {
    const f32x4 gv = ld4(dt + 4 * e), xv = ld4(x + 4 * e), pv = ld4(p + 4 * e), zv = ld4(z + 4 * e);
    f32x4 ox, op;
    if (accumulate_dx) ox = ld4(dx + 4 * e);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gain = 1.f + (pv[j] - mn) / fd;
      const float vx = gv[j] * gain;
      ox[j] = accumulate_dx ? ox[j] + vx : vx;
      float v = gv[j] * xv[j] / fd;
      if (pv[j] == mn) v += gmn;
      if (pv[j] == mx) v += gmx;
      op[j] = v;
      const float vm = (bn_pre(a4[j], zv[j], s4[j], d4[j]) > 0.f) ? v : 0.f;
      s1[j] += (double)vm;
      s2[j] += (double)vm * (double)zv[j];
    }
    *reinterpret_cast<f32x4*>(dx + 4 * e) = ox;
    *reinterpret_cast<f32x4*>(dp + 4 * e) = op;
  }
  col_stats_flush(s1, s2, CV, colpart, lds);
}

// ds = A g + E (s - S) + D (backward apply of a BatchNorm WITHOUT ReLU) with the column sums the backward of the BatchNorm +
// ReLU in front of it needs: ds' = ds [fA (z - fS) + fD > 0]: (sum ds', sum ds' z)
__global__ __launch_bounds__(256) void bn_bwd_next_stats_kernel(const float* __restrict__ g, const float* __restrict__ sx,
                                                                const float* __restrict__ A, const float* __restrict__ E,
                                                                const float* __restrict__ D, const float* __restrict__ S,
                                                                const float* __restrict__ z, const float* __restrict__ fA,
                                                                const float* __restrict__ fD, const float* __restrict__ fS,
                                                                float* __restrict__ ds, long n4, int C, double* __restrict__ colpart) {
  __shared__ double lds[256 * 8];
  const int CV = C >> 2;
  const long G = (long)gridDim.x * 256, e0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c0 = (int)(e0 % CV) * 4;
  const f32x4 a4 = ld4(A + c0), e4 = ld4(E + c0), d4 = ld4(D + c0), s4 = ld4(S + c0);
  const f32x4 fa4 = ld4(fA + c0), fd4 = ld4(fD + c0), fs4 = ld4(fS + c0);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (long e = e0; e < n4; e += G) {
    const f32x4 gv = ld4(g + 4 * e), xv = ld4(sx + 4 * e), zv = ld4(z + 4 * e);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = a4[j] * gv[j];
      v += e4[j] * (xv[j] - s4[j]);
      v += d4[j];
      o[j] = v;
      const float vm = (bn_pre(fa4[j], zv[j], fs4[j], fd4[j]) > 0.f) ? v : 0.f;
      s1[j] += (double)vm;
      s2[j] += (double)vm * (double)zv[j];
    }
    *reinterpret_cast<f32x4*>(ds + 4 * e) = o;
  }
  col_stats_flush(s1, s2, CV, colpart, lds);
}

// workgroups: ~4 float4 per thread, at most 4096 (+ rounding), a multiple of CV / gcd(CV, 256) so that (workgroups * 256) % CV == 0
int fusion_grid(long n4, int CV, long cap = 4096) {
  int g = 256, a = CV;
  while (a) { const int t = g % a; g = a; a = t; }       // gcd(256, CV)
  const int m = CV / g;
  long nb = vr_cdiv(n4, 256 * 2);          // ~2 float4 per thread (16 workgroups per CU at the large maps: the loads of many
  if (nb > cap) nb = cap;                  // waves hide each other's latency; a 512-workgroup grid measured slower than the
                                           // unfused kernels it replaces)
  if (nb < 1) nb = 1;
  nb = vr_cdiv(nb, m) * m;
  return (int)nb;
}
bool fusion_ok(long n, int C) { return C % 4 == 0 && C >= 4 && C <= 1024 && n % C == 0 && n / 4 < (1L << 40); }
#define FUSION_ALIGNED(...)                                                        \
  [&] {                                                                            \
    const void* ps_[] = {__VA_ARGS__};                                             \
    for (const void* q_ : ps_)                                                     \
      if (q_ && !vr_aligned16(q_)) return false;                                   \
    return true;                                                                   \
  }()

}  // namespace

/* Workgroups (= chunk count of the column partials / entries of the (min, max) and four-sum partials) the kernels below use
 * for a contiguous tensor of n elements and C channels; 0: shape not supported (C % 4, C <= 1024).  Buffers: colpart
 * [chunks][C][2] fp64, mmpart [chunks][2] fp32, sums4 [chunks][4] fp64. */
extern "C" int vrnet_fusion_chunks(long n, int C) { return fusion_ok(n, C) ? fusion_grid(n / 4, C / 4) : 0; }
/* ... and of the kernels whose partials EVERY workgroup of the next kernel folds again (mmpart of vrnet_bn_relu_minmax_f32, sums4
 * of vrnet_bn_bwd_enhance_f32): at most 1024, so that the fold stays a few KB per workgroup. */
extern "C" int vrnet_fusion_fold_chunks(long n, int C) { return fusion_ok(n, C) ? fusion_grid(n / 4, C / 4, 1024) : 0; }

/* p = ReLU(A (z - S) + D) (BatchNorm apply + ReLU: normal_conv.py:45-49 inside vr_coc.py:308) with the per-workgroup (min, max)
 * of p for data_normal (vr_coc.py:59-67): vrnet_affine_f32 + the first half of vrnet_minmax_f32 in one launch. */
extern "C" int vrnet_bn_relu_minmax_f32(const float* z, const float* A, const float* D, const float* S, float* p, long n, int C,
                                        float* mmpart, void* stream) {
  VR_CHECK_ARG(z && A && D && S && p && mmpart && fusion_ok(n, C) && FUSION_ALIGNED(z, A, D, S, p), "bn_relu_minmax: bad arguments");
  if (vr_ablated("affine")) return VR_OK;
  hipLaunchKernelGGL((bn_relu_kernel<false>), dim3(fusion_grid(n / 4, C / 4, 1024)), dim3(256), 0, vr_stream(stream), z, A, D, S,
                     (const float*)nullptr, p, n / 4, C, mmpart, (double*)nullptr);
  VR_LAUNCH_CHECK("bn_relu_minmax");
  return VR_OK;
}

/* s = ReLU(A (z - S) + D) + res (vr_coc.py:355-357: inverse_projection's BatchNorm + ReLU, + radar_map) with the column
 * (sum, sumsq) partials of s for the BatchNorm behind it (vrnet_bn_coef_fwd_from_chunks). */
extern "C" int vrnet_bn_relu_res_stats_f32(const float* z, const float* A, const float* D, const float* S, const float* res,
                                           float* s, long n, int C, double* colpart, void* stream) {
  VR_CHECK_ARG(z && A && D && S && res && s && colpart && fusion_ok(n, C) && FUSION_ALIGNED(z, A, D, S, res, s),
               "bn_relu_res_stats: bad arguments");
  if (vr_ablated("affine")) return VR_OK;
  hipLaunchKernelGGL((bn_relu_kernel<true>), dim3(fusion_grid(n / 4, C / 4)), dim3(256), 0, vr_stream(stream), z, A, D, S, res, s,
                     n / 4, C, (float*)nullptr, colpart);
  VR_LAUNCH_CHECK("bn_relu_res_stats");
  return VR_OK;
}

/* t = (1 + data_normal(p)) * x (vr_coc.py:314) from the (min, max) partials of vrnet_bn_relu_minmax_f32 (mm receives the
 * folded pair for the backward pass) with the column (sum, sumsq) partials of t for the BatchNorm behind it (:315). */
extern "C" int vrnet_enhance_stats_f32(const float* p, const float* x, const float* mmpart, int nmm, float* mm, float* t, long n,
                                       int C, double* colpart, void* stream) {
  VR_CHECK_ARG(p && x && mmpart && nmm > 0 && mm && t && colpart && fusion_ok(n, C) && FUSION_ALIGNED(p, x, t),
               "enhance_stats: bad arguments");
  if (vr_ablated("misc")) return VR_OK;
  hipLaunchKernelGGL(enhance_stats_kernel, dim3(fusion_grid(n / 4, C / 4)), dim3(256), 0, vr_stream(stream), p, x, mmpart, nmm, mm,
                     t, n / 4, C, colpart);
  VR_LAUNCH_CHECK("enhance_stats");
  return VR_OK;
}

/* Backward of BatchNorm `norm` (apply: dt = A g + E (t - S) + D) fused with the reduction pass of the gain's backward
 * (vrnet_enhance_bwd_f32's first launch): sums4 [chunks][4]. */
extern "C" int vrnet_bn_bwd_enhance_f32(const float* g, const float* t, const float* A, const float* E, const float* D,
                                        const float* S, const float* x, const float* p, const float* mm, float* dt, long n, int C,
                                        double* sums4, void* stream) {
  VR_CHECK_ARG(g && t && A && E && D && S && x && p && mm && dt && sums4 && fusion_ok(n, C) &&
                   FUSION_ALIGNED(g, t, A, E, D, S, x, p, dt), "bn_bwd_enhance: bad arguments");
  if (vr_ablated("affine")) return VR_OK;
  hipLaunchKernelGGL(bn_bwd_enhance_kernel, dim3(fusion_grid(n / 4, C / 4, 1024)), dim3(256), 0, vr_stream(stream), g, t, A, E, D, S, x, p,
                     mm, dt, n / 4, C, sums4);
  VR_LAUNCH_CHECK("bn_bwd_enhance");
  return VR_OK;
}

/* Second launch of the gain's backward (dx (+)=, dp) fused with the moments pass of BatchNorm bn1's backward: colpart holds the
 * column (sum dp', sum dp' z) of dp' = dp masked by the ReLU recomputed from z with the forward coefficients (fA, fD, fS). */
extern "C" int vrnet_enhance_bwd_stats_f32(const float* dt, const float* x, const float* p, const float* mm, const double* sums4,
                                           int nsums, const float* z, const float* fA, const float* fD, const float* fS, float* dx,
                                           float* dp, long n, int C, int accumulate_dx, double* colpart, void* stream) {
  VR_CHECK_ARG(dt && x && p && mm && sums4 && nsums > 0 && z && fA && fD && fS && dx && dp && colpart && fusion_ok(n, C) &&
                   FUSION_ALIGNED(dt, x, p, z, fA, fD, fS, dx, dp), "enhance_bwd_stats: bad arguments");
  if (vr_ablated("misc")) return VR_OK;
  hipLaunchKernelGGL(enhance_bwd_stats_kernel, dim3(fusion_grid(n / 4, C / 4)), dim3(256), 0, vr_stream(stream), dt, x, p, mm, sums4,
                     nsums, z, fA, fD, fS, dx, dp, n / 4, C, accumulate_dx, colpart);
  VR_LAUNCH_CHECK("enhance_bwd_stats");
  return VR_OK;
}

/* Backward apply of BatchNorm `norm` of RadarEnhanceByImage (ds = A g + E (s - S) + D, vr_coc.py:357) fused with the moments
 * pass of the BatchNorm + ReLU in front of it (inverse_projection's, :355): colpart = column (sum ds', sum ds' z), ds' = ds
 * masked by the ReLU recomputed from z with the forward coefficients (fA, fD, fS). */
extern "C" int vrnet_bn_bwd_next_stats_f32(const float* g, const float* s, const float* A, const float* E, const float* D,
                                           const float* S, const float* z, const float* fA, const float* fD, const float* fS,
                                           float* ds, long n, int C, double* colpart, void* stream) {
  VR_CHECK_ARG(g && s && A && E && D && S && z && fA && fD && fS && ds && colpart && fusion_ok(n, C) &&
                   FUSION_ALIGNED(g, s, A, E, D, S, z, fA, fD, fS, ds), "bn_bwd_next_stats: bad arguments");
  if (vr_ablated("affine")) return VR_OK;
  hipLaunchKernelGGL(bn_bwd_next_stats_kernel, dim3(fusion_grid(n / 4, C / 4)), dim3(256), 0, vr_stream(stream), g, s, A, E, D, S, z,
                     fA, fD, fS, ds, n / 4, C, colpart);
  VR_LAUNCH_CHECK("bn_bwd_next_stats");
  return VR_OK;
}
