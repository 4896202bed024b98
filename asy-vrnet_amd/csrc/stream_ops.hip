// HBM-bound streaming kernels over NHWC fp32 tensors viewed as [B][HW][C] rows with a row stride:
//   * moments: per-(sample, channel) sums in fp64 (feeds GroupNorm, BatchNorm, ShuffleAttention, ECA,
//     layer-scale and bias gradients);
//   * affine: out = pre(A*x1 + D1) + E*x2 + D2 with per-channel or per-(sample, channel) coefficients
//     (normalisation apply forward AND backward, ECA gating, ReLU masks, broadcasts);
//   * the tiny coefficient kernels that turn moments into those coefficients and parameter gradients;
//   * layout copies (channel-strided cat / shuffle, NCHW <-> NHWC), add, fill.
// Reference semantics: GroupNorm(1,C) backbone/fusion/vr_coc.py:105-111; BatchNorm2d in BaseConv
// backbone/conv_utils/normal_conv.py:45; eca backbone/attention_modules/eca.py:16-22; layer scale
// vr_coc.py:264-271.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p);
__device__ __forceinline__ double block_sum(double v, double* red);
// The pre-activation of a normalisation apply, A (x - S) + D, as ONE fused multiply-add everywhere it is evaluated: the
// backward kernels that recompute a ReLU mask from z get the forward's bits.
__device__ __forceinline__ float bn_pre(float A, float x, float S, float D) { return __builtin_fmaf(A, x - S, D); }

// ------------------------------------------------------------------------------------------ moments
template <int VEC>
__global__ __launch_bounds__(256) void moments_kernel(const float* x, long ldx, const float* x2, long ldx2,
                                                      const float* mask, long ldm, long HW, int C, int TPR,
                                                      long rows_per_chunk, int nchunks, double* partial, int total_only,
                                                      const float* gamma = nullptr, double* gtot = nullptr,
                                                      const float* mA = nullptr, const float* mD = nullptr,
                                                      const float* mS = nullptr) {
  extern __shared__ double sm[];   // [256][2*VEC]
  const int tid = threadIdx.x;
  const int tx = tid % TPR, ty = tid / TPR, RP = 256 / TPR;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int CV = C / VEC;
  const int cv = blockIdx.z * TPR + tx;
  const long r0 = chunk * rows_per_chunk;
  const long r1 = min(HW, r0 + rows_per_chunk);
  double s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0;
  if (cv < CV) {
    const long base = (long)b * HW;
    // mA: the ReLU mask is not read but recomputed from x2 = z with the forward coefficients -- bn_pre() is the expression
    // the forward apply evaluated, so the bits agree
    float fa[VEC], fd[VEC], fs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      fa[j] = mA ? mA[cv * VEC + j] : 0.f;
      fd[j] = mA ? mD[cv * VEC + j] : 0.f;
      fs[j] = mA ? mS[cv * VEC + j] : 0.f;
    }
    auto accum = [&](const float (&a)[VEC], const float (&c2)[VEC], const float (&mk)[VEC]) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = a[j];
        if (mask && !(mk[j] > 0.f)) v = 0.f;
        if (mA && !(bn_pre(fa[j], c2[j], fs[j], fd[j]) > 0.f)) v = 0.f;
        s1[j] += (double)v;
        s2[j] += (double)v * (double)(x2 ? c2[j] : v);
      }
    };
    auto load = [&](long r, float (&a)[VEC], float (&c2)[VEC], float (&mk)[VEC]) {
      if (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (base + r) * ldx + cv * 4);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = v[j];
        if (x2) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(x2 + (base + r) * ldx2 + cv * 4);
#pragma unroll
          for (int j = 0; j < VEC; ++j) c2[j] = w[j];
        }
        if (mask) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(mask + (base + r) * ldm + cv * 4);
#pragma unroll
          for (int j = 0; j < VEC; ++j) mk[j] = w[j];
        }
      } else {
        a[0] = x[(base + r) * ldx + cv];
        if (x2) c2[0] = x2[(base + r) * ldx2 + cv];
        if (mask) mk[0] = mask[(base + r) * ldm + cv];
      }
    };
    long r = r0 + ty;
    // four rows per trip: all their loads are issued before the first fp64 accumulate
    for (; r + 3L * RP < r1; r += 4L * RP) {
      float a[4][VEC], c2[4][VEC], mk[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) load(r + (long)u * RP, a[u], c2[u], mk[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(a[u], c2[u], mk[u]);
    }
    for (; r < r1; r += RP) {
      float a[VEC], c2[VEC], mk[VEC];
      load(r, a, c2, mk);
      accum(a, c2, mk);
    }
  }
  if (total_only) {       // GroupNorm(1, C) forward: only the per-sample totals are needed -> one pair per workgroup
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { t1 += s1[j]; t2 += s2[j]; }
    t1 = wave_sum(t1);
    t2 = wave_sum(t2);
    if ((tid & 63) == 0) { sm[2 * (tid >> 6)] = t1; sm[2 * (tid >> 6) + 1] = t2; }
    __syncthreads();
    if (tid < 2)
      partial[(((long)b * nchunks + chunk) * gridDim.z + blockIdx.z) * 2 + tid] = sm[tid] + sm[2 + tid] + sm[4 + tid] + sm[6 + tid];
    return;
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sm[(long)tid * 2 * VEC + j] = s1[j];
    sm[(long)tid * 2 * VEC + VEC + j] = s2[j];
  }
  __syncthreads();
  if (ty == 0 && cv < CV) {
    for (int q = 1; q < RP; ++q) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        s1[j] += sm[(long)(q * TPR + tx) * 2 * VEC + j];
        s2[j] += sm[(long)(q * TPR + tx) * 2 * VEC + VEC + j];
      }
    }
    double* out = partial + (((long)b * nchunks + chunk) * C + (long)cv * VEC) * 2;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      out[2 * j] = s1[j];
      out[2 * j + 1] = s2[j];
    }
  }
  if (gtot) {
    // GroupNorm(1, C) backward: the per-SAMPLE coefficients only need sum_c gamma_c * (s1, s2) -- one pair per workgroup,
    // so that the apply kernel can finish the sample's coefficients itself (vrnet_gn_apply_bwd) without a reduce launch
    double g1 = 0.0, g2 = 0.0;
    if (ty == 0 && cv < CV) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const double g = (double)gamma[cv * VEC + j];
        g1 += g * s1[j];
        g2 += g * s2[j];
      }
    }
    __syncthreads();                      // the channel partials in sm have been consumed
    g1 = wave_sum(g1);
    g2 = wave_sum(g2);
    if ((tid & 63) == 0) { sm[2 * (tid >> 6)] = g1; sm[2 * (tid >> 6) + 1] = g2; }
    __syncthreads();
    if (tid < 2)
      gtot[(((long)b * nchunks + chunk) * gridDim.z + blockIdx.z) * 2 + tid] = sm[tid] + sm[2 + tid] + sm[4 + tid] + sm[6 + tid];
  }
}

// ---- GroupNorm(1, C) with the coefficient step inside the apply kernel (vr_coc.py:105-111) ---------------------------
// Forward: the producing conv's epilogue left (sum, sum of squares) pairs per 32 x 32 tile; every workgroup adds the pairs
// of ITS sample (<= a few thousand, fp64, fixed order: all workgroups of a sample get the same bits) and normalises its
// share of the rows: y = (rstd * gamma) * (x - mean) + beta.  One launch where the coefficient kernel + affine were two.
__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const float* x, long ldx, const double* pairs, long per,
                                                           const float* gamma, const float* beta, float eps, long HW, int C,
                                                           float* y, long ldy, float* mean_rstd, vrnet_planes_out yp) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const double* src = pairs + (long)b * per * 2;
  double s1 = 0, s2 = 0, t1 = 0, t2 = 0;
  long i = threadIdx.x;
  for (; i + 256 < per; i += 512) {
    s1 += src[2 * i]; s2 += src[2 * i + 1];
    t1 += src[2 * (i + 256)]; t2 += src[2 * (i + 256) + 1];
  }
  if (i < per) { s1 += src[2 * i]; s2 += src[2 * i + 1]; }
  s1 = block_sum(s1 + t1, red);
  s2 = block_sum(s2 + t2, red);
  const double n = (double)HW * C;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    mean_rstd[2 * b] = (float)mean;
    mean_rstd[2 * b + 1] = (float)rstd;
  }
  const float mu = (float)mean;
  const int CV = C / 4;
  const long total = HW * CV, stride = (long)gridDim.x * 256;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const long r = e / CV;
    const int c = (int)(e - r * CV) * 4;
    const long row = (long)b * HW + r;
    const f32x4 v = ld4(x + row * ldx + c), g = ld4(gamma + c), be = ld4(beta + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf((float)(rstd * (double)g[j]), v[j] - mu, be[j]);
    if (y) *reinterpret_cast<f32x4*>(y + row * ldy + c) = o;
    if (yp.p) vr_store_planes4(reinterpret_cast<unsigned short*>(yp.p) + row * yp.ld + c, yp.plane, yp.np, o);
  }
}

// Backward: blockIdx.y < B -- dx = A * dy + E * (x - mean) + D (+ add) with the sample's coefficients finished here from
// the moments kernel's gamma-weighted chunk totals; blockIdx.y == B -- the per-channel parameter gradients (one wave per
// channel over the [B][nchunks] chunk partials).  One launch where reduce + coefficient kernel + affine were three.
__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                           const double* gtot, int gper, const double* partial, int nchunks,
                                                           const float* mean_rstd, const float* gamma, int B, long HW, int C,
                                                           const float* add, long ldadd, float* out, long ldo, float* dgamma,
                                                           float* dbeta, int accumulate_params, vrnet_planes_out outp) {
  __shared__ double red[4];
  if ((int)blockIdx.y == B) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    const int total = B * nchunks;
    for (int j = lane; j < total; j += 64) {
      const int bb = j / nchunks;
      const double mu = mean_rstd[2 * bb], r = mean_rstd[2 * bb + 1];
      const double* q = partial + ((long)j * C + c) * 2;
      const double s1 = q[0], s2 = q[1];
      dg += r * (s2 - mu * s1);
      db += s1;
    }
    dg = wave_sum(dg);
    db = wave_sum(db);
    if (lane == 0) {
      dgamma[c] = (accumulate_params ? dgamma[c] : 0.f) + (float)dg;
      dbeta[c] = (accumulate_params ? dbeta[c] : 0.f) + (float)db;
    }
    return;
  }
  const int b = blockIdx.y;
  const double* src = gtot + (long)b * gper * 2;
  double t1 = 0, t2 = 0;
  for (int i = threadIdx.x; i < gper; i += 256) { t1 += src[2 * i]; t2 += src[2 * i + 1]; }
  t1 = block_sum(t1, red);
  t2 = block_sum(t2, red);
  const double mu = mean_rstd[2 * b], r = mean_rstd[2 * b + 1];
  const double n = (double)HW * C;
  const double m1 = t1 / n, m2 = r * (t2 - mu * t1) / n;
  const float e_ = (float)(-r * r * m2), d_ = (float)(-r * m1), muf = (float)mu;
  const int CV = C / 4;
  const long total = HW * CV, stride = (long)gridDim.x * 256;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const long rr = e / CV;
    const int c = (int)(e - rr * CV) * 4;
    const long row = (long)b * HW + rr;
    const f32x4 g = ld4(dy + row * lddy + c), xv = ld4(x + row * ldx + c), ga = ld4(gamma + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaf((float)(r * (double)ga[j]), g[j], fmaf(e_, xv[j] - muf, d_));
    if (add) {
      const f32x4 a = ld4(add + row * ldadd + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += a[j];
    }
    *reinterpret_cast<f32x4*>(out + row * ldo + c) = o;
    if (outp.p) vr_store_planes4(reinterpret_cast<unsigned short*>(outp.p) + row * outp.ld + c, outp.plane, outp.np, o);
  }
}

// out[b][c][w] = sum_k partial[b][k][c][w].  16 lanes share one output (lane l adds chunks l, l + 16, ...; the 16
// partial sums are then added in lane order through LDS: fixed order, deterministic) -- a one-thread-per-output
// loop over up to 512 chunks is a serial chain of dependent-latency loads on the critical path of every norm.
__global__ __launch_bounds__(256) void moments_reduce_kernel(const double* partial, double* out, int B, int nchunks, int C) {
  __shared__ double red[16][16];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + o;   // over B*C*2
  const bool live = e < (long)B * C * 2;
  double s = 0.0;
  if (live) {
    const long b = e / (2L * C), rem = e - b * 2L * C;
    for (int k = sl; k < nchunks; k += 16) s += partial[((long)b * nchunks + k) * C * 2 + rem];
  }
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && live) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][o];
    out[e] = s;
  }
}

// ------------------------------------------------------------------------------------------ affine
struct AffineArgs {
  const float* x1; long ld1; const float* A; const float* D1; const float* S1;
  const float* masky; long ldm;
  const float* x2; long ld2; const float* E; const float* D2; const float* S2;
  float* out; long ldo;
  long HW; int C; long bstride; int pre; int accumulate;
  const float* add; long ldadd;      // plain addend (out-of-place accumulate); accumulate = 1 is add == out
  const float* mA; const float* mD; const float* mS;   // pre == 3: ReLU mask = [mA (x2 - mS) + mD > 0], per channel
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// One element (VEC consecutive channels of one pixel).  Every operand -- coefficients too -- moves as one 16-byte
// access on the vector path (C % 4 == 0 keeps cb + c a multiple of 4).
template <int VEC>
__device__ __forceinline__ void affine_one(const AffineArgs& p, long row, long cb, int c) {
  if (VEC == 4) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
    f32x4 v = p.D1 ? ld4(p.D1 + cb + c) : zero;
    // issue the streaming loads first: they are independent and the longest-latency operands
    f32x4 a = zero, x2 = zero, mk = zero, acc = zero;
    if (p.x1) a = ld4(p.x1 + row * p.ld1 + c);
    if (p.x2) x2 = ld4(p.x2 + row * p.ld2 + c);
    if (p.pre == 2) mk = ld4(p.masky + row * p.ldm + c);
    float* o = p.out + row * p.ldo + c;
    if (p.add) acc = ld4(p.add + row * p.ldadd + c);
    if (p.x1) {
      const f32x4 Aq = p.A ? ld4(p.A + cb + c) : one, Sq = p.S1 ? ld4(p.S1 + cb + c) : zero;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = bn_pre(Aq[j], a[j], Sq[j], v[j]);
    }
    if (p.pre == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    } else if (p.pre == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(mk[j] > 0.f)) v[j] = 0.f;
    } else if (p.pre == 3) {
      const f32x4 fa = ld4(p.mA + c), fd = ld4(p.mD + c), fs = ld4(p.mS + c);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(bn_pre(fa[j], x2[j], fs[j], fd[j]) > 0.f)) v[j] = 0.f;
    }
    if (p.x2) v += (p.E ? ld4(p.E + cb + c) : one) * (x2 - (p.S2 ? ld4(p.S2 + cb + c) : zero));
    if (p.D2) v += ld4(p.D2 + cb + c);
    if (p.add) v += acc;
    *reinterpret_cast<f32x4*>(o) = v;
  } else {
    float v = p.D1 ? p.D1[cb + c] : 0.f;
    if (p.x1) v = bn_pre(p.A ? p.A[cb + c] : 1.f, p.x1[row * p.ld1 + c], p.S1 ? p.S1[cb + c] : 0.f, v);
    if (p.pre == 1) v = fmaxf(v, 0.f);
    else if (p.pre == 2 && !(p.masky[row * p.ldm + c] > 0.f)) v = 0.f;
    else if (p.pre == 3 && !(bn_pre(p.mA[c], p.x2[row * p.ld2 + c], p.mS[c], p.mD[c]) > 0.f)) v = 0.f;
    if (p.x2) v += (p.E ? p.E[cb + c] : 1.f) * (p.x2[row * p.ld2 + c] - (p.S2 ? p.S2[cb + c] : 0.f));
    if (p.D2) v += p.D2[cb + c];
    float* o = p.out + row * p.ldo + c;
    if (p.add) v += p.add[row * p.ldadd + c];
    o[0] = v;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void affine_kernel(const AffineArgs p) {
  const int CV = p.C / VEC;
  const long total = p.HW * CV;
  const long b = blockIdx.y;
  const long cb = b * p.bstride;
  const long stride = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  // two elements per trip: both sets of loads are in flight before the first store
  for (; e + stride < total; e += 2 * stride) {
    const long r0 = e / CV, r1 = (e + stride) / CV;
    affine_one<VEC>(p, b * p.HW + r0, cb, (int)(e - r0 * CV) * VEC);
    affine_one<VEC>(p, b * p.HW + r1, cb, (int)(e + stride - r1 * CV) * VEC);
  }
  if (e < total) {
    const long r0 = e / CV;
    affine_one<VEC>(p, b * p.HW + r0, cb, (int)(e - r0 * CV) * VEC);
  }
}

// The same for channel counts that are no multiple of 4 (the 3 / 4 / 7-channel maps of the input fusion at FULL resolution:
// 2 M pixels per sample set, vr_coc.py:303-359) when every tensor is contiguous (row stride == C): a sample is then one flat run
// of HW * C floats, read and written 16 bytes at a time; the channel of element i is i mod C (round 5: the scalar path above ran
// these launches -- all of them on the step's critical chain -- at 2 TB/s with a 64-bit division per element).
__global__ __launch_bounds__(256) void affine_flat_kernel(const AffineArgs p) {
  const long n4 = p.HW * p.C / 4;
  const long b = blockIdx.y, cb = b * p.bstride, base = b * p.HW * p.C;
  const int C = p.C;
  auto one = [&](long e) {
    const long i0 = 4 * e;
    int c = (int)(i0 % C);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, x2 = a, mk = a, acc = a, v;
    if (p.x1) a = ld4(p.x1 + base + i0);
    if (p.x2) x2 = ld4(p.x2 + base + i0);
    if (p.pre == 2) mk = ld4(p.masky + base + i0);
    if (p.add) acc = ld4(p.add + base + i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = p.D1 ? p.D1[cb + c] : 0.f;
      if (p.x1) t = bn_pre(p.A ? p.A[cb + c] : 1.f, a[j], p.S1 ? p.S1[cb + c] : 0.f, t);
      if (p.pre == 1) t = fmaxf(t, 0.f);
      else if (p.pre == 2 && !(mk[j] > 0.f)) t = 0.f;
      else if (p.pre == 3 && !(bn_pre(p.mA[c], x2[j], p.mS[c], p.mD[c]) > 0.f)) t = 0.f;
      if (p.x2) t += (p.E ? p.E[cb + c] : 1.f) * (x2[j] - (p.S2 ? p.S2[cb + c] : 0.f));
      if (p.D2) t += p.D2[cb + c];
      if (p.add) t += acc[j];
      v[j] = t;
      if (++c == C) c = 0;
    }
    *reinterpret_cast<f32x4*>(p.out + base + i0) = v;
  };
  const long stride = (long)gridDim.x * 256;
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  for (; e + stride < n4; e += 2 * stride) {
    one(e);
    one(e + stride);
  }
  if (e < n4) one(e);
}

// ------------------------------------------------------------------------------------------ coefficients
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

__global__ __launch_bounds__(256) void gn_coef_fwd_kernel(const double* mom, const float* gamma, const float* beta,
                                                          float eps, long HW, int C, float* A, float* D,
                                                          float* S, float* mean_rstd, const float* gamma2,
                                                          const float* beta2) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  if (gamma2 && 2 * b >= (int)gridDim.x) { gamma = gamma2; beta = beta2; }     // two-stream launch: second half of the samples
  double s1 = 0, s2 = 0;
  for (int c = threadIdx.x; c < C; c += 256) {
    s1 += mom[((long)b * C + c) * 2];
    s2 += mom[((long)b * C + c) * 2 + 1];
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const double n = (double)HW * C;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int c = threadIdx.x; c < C; c += 256) {
    A[(long)b * C + c] = (float)(rstd * gamma[c]);      // y = A * (x - S) + D: the subtraction comes first, as in
    D[(long)b * C + c] = beta[c];                         // torch (A*x + (beta - mean*A) cancels when |mean| >> std)
    S[(long)b * C + c] = (float)mean;
  }
  if (threadIdx.x == 0) {
    mean_rstd[2 * b] = (float)mean;
    mean_rstd[2 * b + 1] = (float)rstd;
  }
}

// GroupNorm(1, C) forward statistics straight from the moments kernel: in `total_only` mode every workgroup of the
// moments kernel emits one (sum, sum of squares) pair, and this kernel adds the <= 512 pairs of a sample and writes
// the coefficients -- the cross-chunk reduce launch and the per-channel table are skipped.
__global__ __launch_bounds__(256) void gn_coef_fwd_partial_kernel(const double* partial, long pairs, const float* gamma,
                                                                  const float* beta, float eps, long HW, int C, float* A,
                                                                  float* D, float* S, float* mean_rstd,
                                                                  const float* gamma2, const float* beta2) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  if (gamma2 && 2 * b >= (int)gridDim.x) { gamma = gamma2; beta = beta2; }     // two-stream launch: second half of the samples
  const double* src = partial + (long)b * pairs * 2;
  double s1 = 0, s2 = 0, t1 = 0, t2 = 0;
  long i = threadIdx.x;
  for (; i + 256 < pairs; i += 512) {          // two independent pairs per trip
    s1 += src[2 * i]; s2 += src[2 * i + 1];
    t1 += src[2 * (i + 256)]; t2 += src[2 * (i + 256) + 1];
  }
  if (i < pairs) { s1 += src[2 * i]; s2 += src[2 * i + 1]; }
  s1 = block_sum(s1 + t1, red);
  s2 = block_sum(s2 + t2, red);
  const double n = (double)HW * C;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int c = threadIdx.x; c < C; c += 256) {
    A[(long)b * C + c] = (float)(rstd * gamma[c]);
    D[(long)b * C + c] = beta[c];
    S[(long)b * C + c] = (float)mean;
  }
  if (threadIdx.x == 0) {
    mean_rstd[2 * b] = (float)mean;
    mean_rstd[2 * b + 1] = (float)rstd;
  }
}

// blocks [0, B): per-sample dx coefficients; blocks [B, ...): per-channel dgamma / dbeta
__global__ __launch_bounds__(256) void gn_coef_bwd_kernel(const double* mom2, const float* mean_rstd,
                                                          const float* gamma, int B, long HW, int C, float* A,
                                                          float* E, float* D, float* S, float* dgamma, float* dbeta,
                                                          int accumulate, const float* gamma2, float* dgamma2,
                                                          float* dbeta2) {
  __shared__ double red[4];
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
    if (gamma2 && 2 * b >= B) gamma = gamma2;
    const double mu = mean_rstd[2 * b], r = mean_rstd[2 * b + 1];
    double t1 = 0, t2 = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
      const double g = gamma[c], s1 = mom2[((long)b * C + c) * 2], s2 = mom2[((long)b * C + c) * 2 + 1];
      t1 += g * s1;
      t2 += g * (s2 - mu * s1);
    }
    t1 = block_sum(t1, red);
    t2 = block_sum(t2, red);
    const double n = (double)HW * C;
    const double m1 = t1 / n, m2 = r * t2 / n;
    const float e = (float)(-r * r * m2), d = (float)(-r * m1);     // dx = A*dy + E*(x - S) + D
    for (int c = threadIdx.x; c < C; c += 256) {
      A[(long)b * C + c] = (float)(r * gamma[c]);
      E[(long)b * C + c] = e;
      D[(long)b * C + c] = d;
      S[(long)b * C + c] = (float)mu;
    }
  } else {
    const int c = (blockIdx.x - B) * 256 + threadIdx.x;
    if (c >= C) return;
    const int nstream = gamma2 ? 2 : 1, per = B / nstream;
    for (int z = 0; z < nstream; ++z) {       // per-channel gradients of each stream's own parameters
      double dg = 0, db = 0;
      for (int b = z * per; b < (z + 1) * per; ++b) {
        const double mu = mean_rstd[2 * b], r = mean_rstd[2 * b + 1];
        const double s1 = mom2[((long)b * C + c) * 2], s2 = mom2[((long)b * C + c) * 2 + 1];
        dg += r * (s2 - mu * s1);
        db += s1;
      }
      float* og = z ? dgamma2 : dgamma;
      float* ob = z ? dbeta2 : dbeta;
      og[c] = (accumulate ? og[c] : 0.f) + (float)dg;
      ob[c] = (accumulate ? ob[c] : 0.f) + (float)db;
    }
  }
}

__global__ void bn_coef_fwd_kernel(const double* mom, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, long long* nbt,
                                   int training, int B, long HW, int C, float* A, float* D, float* S,
                                   float* mean_rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && nbt) *nbt += 1;
  if (c >= C) return;
  double mean, var;
  if (training) {
    double s1 = 0, s2 = 0;
    for (int b = 0; b < B; ++b) {
      s1 += mom[((long)b * C + c) * 2];
      s2 += mom[((long)b * C + c) * 2 + 1];
    }
    const double n = (double)B * HW;
    mean = s1 / n;
    var = s2 / n - mean * mean;
    if (var < 0) var = 0;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (n / (n - 1.0)));
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  A[c] = (float)(rstd * gamma[c]);     // y = A * (z - S) + D
  D[c] = beta[c];
  S[c] = (float)mean;
  mean_rstd[2 * c] = (float)mean;
  mean_rstd[2 * c + 1] = (float)rstd;
}

__global__ void bn_coef_bwd_kernel(const double* mom2, const float* mean_rstd, const float* gamma, int training,
                                   int B, long HW, int C, float* A, float* E, float* D, float* S, float* dgamma,
                                   float* dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int b = 0; b < B; ++b) {
    s1 += mom2[((long)b * C + c) * 2];
    s2 += mom2[((long)b * C + c) * 2 + 1];
  }
  const double mu = mean_rstd[2 * c], r = mean_rstd[2 * c + 1], g = gamma[c];
  const double n = (double)B * HW;
  const double dxh = r * (s2 - mu * s1);   // sum dy * xhat
  A[c] = (float)(g * r);
  if (training) {
    const double m1 = s1 / n, m2 = dxh / n;
    E[c] = (float)(-g * r * r * m2);     // dz = A*dy' + E*(z - S) + D
    D[c] = (float)(-g * r * m1);
  } else {
    E[c] = 0.f;
    D[c] = 0.f;
  }
  S[c] = (float)mu;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)dxh;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
}

// BatchNorm coefficients straight from the moments kernel's chunk partials [B][nchunks][C][2]: ONE WAVE per channel -- lane l
// adds partials l, l + 64, ... (four independent loads in flight per trip), the 64 lane sums are added by a butterfly in a
// fixed pattern (deterministic), lane 0 finishes the channel.  The separate moments_reduce launch and the [B][C][2] table
// disappear (100 launches per training step).  (Round 2 gave a channel 16 lanes of a workgroup that covered 16 channels:
// C / 16 workgroups -- 4 at the widest maps -- each walking up to 128 partials per lane in a dependent loop, 18 us per
// launch on average for an O(C) result.)
__device__ __forceinline__ void bn_channel_sums(const double* partial, int B, int nchunks, int C, int c, bool live,
                                                double& s1, double& s2) {
  const int lane = threadIdx.x & 63;
  double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0, c1 = 0.0, c2 = 0.0, d1 = 0.0, d2 = 0.0;
  if (live) {
    const int total = B * nchunks;
    const double* base = partial + (long)c * 2;
    const long step = (long)C * 2;
    int j = lane;
    for (; j + 192 < total; j += 256) {
      const double* q0 = base + (long)j * step;
      const double* q1 = q0 + 64 * step;
      const double* q2 = q1 + 64 * step;
      const double* q3 = q2 + 64 * step;
      const double v0 = q0[0], w0 = q0[1], v1 = q1[0], w1 = q1[1], v2 = q2[0], w2 = q2[1], v3 = q3[0], w3 = q3[1];
      a1 += v0; a2 += w0; b1 += v1; b2 += w1; c1 += v2; c2 += w2; d1 += v3; d2 += w3;
    }
    for (; j < total; j += 64) {
      const double* q0 = base + (long)j * step;
      a1 += q0[0];
      a2 += q0[1];
    }
  }
  s1 = wave_sum((a1 + b1) + (c1 + d1));
  s2 = wave_sum((a2 + b2) + (c2 + d2));
}

__global__ __launch_bounds__(256) void bn_coef_fwd_partial_kernel(const double* partial, int nchunks, const float* gamma,
                                                                  const float* beta, float eps, float momentum,
                                                                  float* running_mean, float* running_var, long long* nbt,
                                                                  int B, long HW, int C, float* A, float* D, float* S,
                                                                  float* mean_rstd) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = c < C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double s1, s2;
  bn_channel_sums(partial, B, nchunks, C, c, live, s1, s2);
  if ((threadIdx.x & 63) != 0 || !live) return;
  const double n = (double)B * HW;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
  running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (n / (n - 1.0)));
  const double rstd = 1.0 / sqrt(var + (double)eps);
  A[c] = (float)(rstd * gamma[c]);     // y = A * (z - S) + D
  D[c] = beta[c];
  S[c] = (float)mean;
  mean_rstd[2 * c] = (float)mean;
  mean_rstd[2 * c + 1] = (float)rstd;
}

__global__ __launch_bounds__(256) void bn_coef_bwd_partial_kernel(const double* partial, int nchunks, const float* mean_rstd,
                                                                  const float* gamma, int training, int B, long HW, int C,
                                                                  float* A, float* E, float* D, float* S, float* dgamma,
                                                                  float* dbeta, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = c < C;
  double s1, s2;
  bn_channel_sums(partial, B, nchunks, C, c, live, s1, s2);
  if ((threadIdx.x & 63) != 0 || !live) return;
  const double mu = mean_rstd[2 * c], r = mean_rstd[2 * c + 1], g = gamma[c];
  const double n = (double)B * HW;
  const double dxh = r * (s2 - mu * s1);   // sum dy * xhat
  A[c] = (float)(g * r);
  if (training) {
    const double m1 = s1 / n, m2 = dxh / n;
    E[c] = (float)(-g * r * r * m2);     // dz = A*dy' + E*(z - S) + D
    D[c] = (float)(-g * r * m1);
  } else {
    E[c] = 0.f;
    D[c] = 0.f;
  }
  S[c] = (float)mu;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)dxh;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
}

// gate[b][c] = sigmoid(sum_j wk[j] * mean[b][c + j - pad]);  eca.py:16-22
__global__ void eca_coef_fwd_kernel(const double* mom, const float* wk, int k, int B, long HW, int C, float* gate) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C) return;
  const int b = e / C, c = e % C, pad = (k - 1) / 2;
  double z = 0;
  for (int j = 0; j < k; ++j) {
    const int cc = c + j - pad;
    if (cc >= 0 && cc < C) z += (double)wk[j] * (mom[((long)b * C + cc) * 2] / (double)HW);
  }
  gate[e] = (float)(1.0 / (1.0 + exp(-z)));
}

// mom2 = (sum dy, sum dy*x) per (b,c); mom = x moments.  F[b][c] = dm[b][c] / HW (constant added to dx).
// dwk (k taps) is reduced by block 0 only after the per-element pass (second kernel below).
__global__ void eca_coef_bwd_kernel(const double* mom2, const float* gate, const float* wk, int k, int B, long HW,
                                    int C, float* F) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C) return;
  const int b = e / C, c = e % C, pad = (k - 1) / 2;
  double dm = 0;
  for (int j = 0; j < k; ++j) {
    const int cc = c - j + pad;   // z[b][cc] uses m[b][cc + j - pad] = m[b][c]
    if (cc >= 0 && cc < C) {
      const double g = gate[(long)b * C + cc];
      dm += (double)wk[j] * mom2[((long)b * C + cc) * 2 + 1] * g * (1.0 - g);
    }
  }
  F[e] = (float)(dm / (double)HW);
}

__global__ __launch_bounds__(256) void eca_dwk_kernel(const double* mom2, const double* mom, const float* gate,
                                                      int k, int B, long HW, int C, float* dwk, int accumulate) {
  __shared__ double red[4];
  const int j = blockIdx.x, pad = (k - 1) / 2;
  double s = 0;
  for (long e = threadIdx.x; e < (long)B * C; e += 256) {
    const int b = e / C, c = e % C, cc = c + j - pad;
    if (cc >= 0 && cc < C) {
      const double g = gate[e];
      s += mom2[e * 2 + 1] * g * (1.0 - g) * (mom[((long)b * C + cc) * 2] / (double)HW);
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) dwk[j] = (accumulate ? dwk[j] : 0.f) + (float)s;
}

// layer scale: x_new = x + ls * o.  mom2 = (sum dx, sum dx*o).  dls = sum_b S2; dbias = ls * sum_b S1.
__global__ void ls_coef_bwd_kernel(const double* mom2, const float* ls, int B, int C, float* dls, float* dbias,
                                   int accumulate, const float* ls2, float* dls2, float* dbias2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int b0 = 0, b1 = B;
  if (gridDim.y > 1) {       // two-stream launch: blockIdx.y = stream, each over its half of the samples
    b0 = blockIdx.y * (B / 2); b1 = b0 + B / 2;
    if (blockIdx.y) { ls = ls2; dls = dls2; dbias = dbias2; }
  }
  double s1 = 0, s2 = 0;
  for (int b = b0; b < b1; ++b) {
    s1 += mom2[((long)b * C + c) * 2];
    s2 += mom2[((long)b * C + c) * 2 + 1];
  }
  if (dls) dls[c] = (accumulate ? dls[c] : 0.f) + (float)s2;
  if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)((ls ? (double)ls[c] : 1.0) * s1);
}

// out[b][c] = mom[b][c][0] * scale   (global average pooling, per-sample column sums)
__global__ void moments_to_float_kernel(const double* mom, float* out, long n, double scale, int which) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = (float)(mom[2 * e + which] * scale);
}

// ------------------------------------------------------------------------------------------ copies
// dst[r*ldd + c*dcs] (+)= src[r*lds + c*scs]
__global__ void copy_channels_kernel(const float* src, long lds, int scs, float* dst, long ldd, int dcs, long rows,
                                     int C, int accumulate) {
  const long total = rows * C;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / C;
    const int c = e - r * C;
    const float v = src[r * lds + (long)c * scs];
    float* d = dst + r * ldd + (long)c * dcs;
    *d = accumulate ? *d + v : v;
  }
}

// torch.cat([a, b], 1) (+ the 2-group channel shuffle when `interleave`: channel 2 j = a_j, 2 j + 1 = b_j) in ONE launch, and
// its adjoint: dir 0: cat[r][c] = a / b;  dir 1: a[r][j] (+)= cat[r][..], b[r][j] (+)= cat[r][..] (per-source accumulate flags;
// a source pointer may be null: that half is skipped).  Consecutive threads walk consecutive channels of the wide tensor.
__global__ void cat2_kernel(float* a, long lda, int Ca, float* b, long ldb, int Cb, float* cat, long ldc, long rows,
                            int interleave, int dir, int acc_a, int acc_b) {
  const int Ct = Ca + Cb;
  const long total = rows * Ct;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / Ct;
    const int c = e - r * Ct;
    const bool second = interleave ? (c & 1) : (c >= Ca);
    const int j = interleave ? (c >> 1) : (second ? c - Ca : c);
    float* s = second ? (b ? b + r * ldb + j : nullptr) : (a ? a + r * lda + j : nullptr);
    if (!s) continue;
    float* w = cat + r * ldc + c;
    if (dir == 0) {
      *w = *s;
    } else {
      const int acc = second ? acc_b : acc_a;
      *s = acc ? *s + *w : *w;
    }
  }
}

// The same on 16-byte accesses (channel counts and row strides % 4, aligned bases): a thread moves four consecutive source
// channels; with the 2-group shuffle it takes the quad of BOTH halves and writes the eight interleaved outputs.  (Round 4: the
// scalar kernel above spent 24 us per launch on a 64-bit division per element; 20 launches per step on the chain.)
__global__ __launch_bounds__(256) void cat2_vec_kernel(float* a, long lda, int Ca, float* b, long ldb, int Cb, float* cat,
                                                       long ldc, long rows, int interleave, int dir, int acc_a, int acc_b) {
  const int qa = Ca >> 2, qb = Cb >> 2;
  const int per_row = interleave ? qa : qa + qb;
  const long total = rows * per_row;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / per_row;
    const int q = (int)(e - r * per_row);
    if (interleave) {
      float* w = cat + r * ldc + 8 * q;
      if (dir == 0) {
        const f32x4 u = ld4(a + r * lda + 4 * q), v = ld4(b + r * ldb + 4 * q);
        const f32x4 lo = {u[0], v[0], u[1], v[1]}, hi = {u[2], v[2], u[3], v[3]};
        *reinterpret_cast<f32x4*>(w) = lo;
        *reinterpret_cast<f32x4*>(w + 4) = hi;
      } else {
        const f32x4 lo = ld4(w), hi = ld4(w + 4);
        if (a) {
          f32x4 u = {lo[0], lo[2], hi[0], hi[2]};
          f32x4* d = reinterpret_cast<f32x4*>(a + r * lda + 4 * q);
          if (acc_a) u += *d;
          *d = u;
        }
        if (b) {
          f32x4 v = {lo[1], lo[3], hi[1], hi[3]};
          f32x4* d = reinterpret_cast<f32x4*>(b + r * ldb + 4 * q);
          if (acc_b) v += *d;
          *d = v;
        }
      }
    } else {
      const bool second = q >= qa;
      float* src = second ? b : a;
      if (!src) continue;
      float* sp = second ? b + r * ldb + 4 * (q - qa) : a + r * lda + 4 * q;
      float* w = cat + r * ldc + (second ? Ca + 4 * (q - qa) : 4 * q);
      if (dir == 0) {
        *reinterpret_cast<f32x4*>(w) = ld4(sp);
      } else {
        f32x4 t = ld4(w);
        if (second ? acc_b : acc_a) t += ld4(sp);
        *reinterpret_cast<f32x4*>(sp) = t;
      }
    }
  }
}

// The 3 + 4 -> 7-channel concat of the input fusion (2 M rows) and its adjoint, one thread per ROW (round 5): the generic kernel
// above does a 64-bit division per element and ran the two launches at 37 us each on the critical chain.
template <int CA, int CB>
__global__ __launch_bounds__(256) void cat2_rows_kernel(float* a, long lda, float* b, long ldb, float* cat, long ldc, long rows,
                                                        int dir, int acc_a, int acc_b, int vecb) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    float* w = cat + r * ldc;
    if (dir == 0) {
      float u[CA], v[CB];
#pragma unroll
      for (int j = 0; j < CA; ++j) u[j] = a[r * lda + j];
      if (CB == 4 && vecb) {
        const f32x4 t = ld4(b + r * ldb);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = t[j];
      } else {
#pragma unroll
        for (int j = 0; j < CB; ++j) v[j] = b[r * ldb + j];
      }
#pragma unroll
      for (int j = 0; j < CA; ++j) w[j] = u[j];
#pragma unroll
      for (int j = 0; j < CB; ++j) w[CA + j] = v[j];
    } else {
      float t[CA + CB];
#pragma unroll
      for (int j = 0; j < CA + CB; ++j) t[j] = w[j];
      if (a) {
#pragma unroll
        for (int j = 0; j < CA; ++j) a[r * lda + j] = acc_a ? a[r * lda + j] + t[j] : t[j];
      }
      if (b) {
        if (CB == 4 && vecb) {
          f32x4 o = {t[CA], t[CA + 1], t[CA + 2], t[CA + 3]};
          if (acc_b) o += ld4(b + r * ldb);
          *reinterpret_cast<f32x4*>(b + r * ldb) = o;
        } else {
#pragma unroll
          for (int j = 0; j < CB; ++j) b[r * ldb + j] = acc_b ? b[r * ldb + j] + t[CA + j] : t[CA + j];
        }
      }
    }
  }
}

// [B][C][HW] -> [B][HW][ld] for C <= 4 (the network's two inputs), one thread per pixel: C coalesced plane reads, one 12- or
// 16-byte row written (round 5: the 32 x 32 tile kernel below leaves 28 of its 32 channel lanes idle at C = 3 / 4: 26 us each)
template <int C>
__global__ __launch_bounds__(256) void nchw_to_nhwc_small_kernel(const float* src, float* dst, long ldd, long HW, int vec) {
  const long b = blockIdx.y;
  for (long pp = (long)blockIdx.x * 256 + threadIdx.x; pp < HW; pp += (long)gridDim.x * 256) {
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = src[(b * C + c) * HW + pp];
    float* d = dst + (b * HW + pp) * ldd;
    if (C == 4 && vec) {
      const f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(d) = o;
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] = v[c];
    }
  }
}

// 32x32 LDS-tiled transpose between [B][C][HW] and [B][HW][ld]
__global__ void nchw_to_nhwc_kernel(const float* src, float* dst, long ldd, int C, long HW) {
  __shared__ float tile[32][33];
  const long b = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;   // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const long pp = p0 + tx;
    tile[i][tx] = (c < C && pp < HW) ? src[(b * C + c) * HW + pp] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const long pp = p0 + i;
    const int c = c0 + tx;
    if (c < C && pp < HW) dst[(b * HW + pp) * ldd + c] = tile[tx][i];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* src, long lds, float* dst, int C, long HW, int accumulate) {
  __shared__ float tile[32][33];
  const long b = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  for (int i = ty; i < 32; i += 8) {
    const long pp = p0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (c < C && pp < HW) ? src[(b * HW + pp) * lds + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const long pp = p0 + tx;
    if (c < C && pp < HW) {
      float* d = dst + (b * C + c) * HW + pp;
      *d = accumulate ? *d + tile[tx][i] : tile[tx][i];
    }
  }
}

__global__ void add_kernel(float* dst, const float* src, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) dst[e] += src[e];
}
__global__ void fill_kernel(float* dst, float v, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) dst[e] = v;
}

// RadarEnhanceByImage (vr_coc.py:343-352), three launches in one (round 5): the ShuffleAttention apply (gate from P, Q, Mn of
// vrnet_sa_coef_fwd, output channels in the module's shuffled order), torch.cat with the radar map + the 2-group channel shuffle,
// and the per-(sample, channel) sums of the result that the ECA gate behind it needs.  Thread (tx = output quad oq of the 2 C
// channels, ty = row): cat[4 oq .. 4 oq + 3] = {SA(x)[oq-th even slot], r[2 oq], SA(x)[odd slot], r[2 oq + 1]}, where the two
// SA values come from input channels oq and C / 2 + oq (the inverse of sa_dst).  Partial sums in the moments layout
// [b][chunk][2 C][2] (the second component, a sum of squares nobody reads, is written as 0).
__global__ __launch_bounds__(256) void sa_cat_sums_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ P,
                                                          const float* __restrict__ Q, const float* __restrict__ Mn,
                                                          const float* __restrict__ r, long ldr, float* __restrict__ cat, long ldc,
                                                          long HW, int C, int TPR, long rows_per_chunk, int nchunks,
                                                          double* __restrict__ partial) {
  extern __shared__ double sm[];   // [256][4]
  const int tid = threadIdx.x;
  const int tx = tid % TPR, ty = tid / TPR, RP = 256 / TPR;
  const int chunk = blockIdx.x, b = blockIdx.y, H2 = C >> 1;
  const long r0 = chunk * rows_per_chunk, r1 = min(HW, r0 + rows_per_chunk);
  double s[4] = {0, 0, 0, 0};
  if (tx < H2) {
    const float p0 = P[b * C + tx], q0 = Q[b * C + tx], m0 = Mn[b * C + tx];
    const float p1 = P[b * C + H2 + tx], q1 = Q[b * C + H2 + tx], m1 = Mn[b * C + H2 + tx];
    for (long rr = r0 + ty; rr < r1; rr += RP) {
      const long row = (long)b * HW + rr;
      const float x0 = x[row * ldx + tx], x1 = x[row * ldx + H2 + tx];
      const float2 rv = *reinterpret_cast<const float2*>(r + row * ldr + 2 * tx);
      const float a0 = x0 * vr_sigmoid(p0 * (x0 - m0) + q0), a1 = x1 * vr_sigmoid(p1 * (x1 - m1) + q1);
      const f32x4 o = {a0, rv.x, a1, rv.y};
      *reinterpret_cast<f32x4*>(cat + row * ldc + 4 * tx) = o;
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (double)o[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[(long)tid * 4 + j] = s[j];
  __syncthreads();
  if (ty == 0 && tx < H2) {
    for (int q = 1; q < RP; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += sm[(long)(q * TPR + tx) * 4 + j];
    double* out = partial + (((long)b * nchunks + chunk) * (2 * C) + 4 * tx) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      out[2 * j] = s[j];
      out[2 * j + 1] = 0.0;
    }
  }
}

static int moments_plan(int B, long HW, int C, int vec, int* TPR, int* ncb, int* nchunks, long* rows) {
  const int CV = C / vec;
  int t = 1;
  while (t < CV && t < 256) t <<= 1;
  *TPR = t;
  *ncb = (int)vr_cdiv(CV, t);
  long nc = vr_cdiv(HW * C, 4096);    // >= 16 elements per thread: B x nc x ncb workgroups must cover 256 CUs several
  if (nc < 1) nc = 1;                  // times over even for one 16x16 map; the chunk reduce is lane-parallel
  if (nc > 512) nc = 512;
  // ... but not more chunks than ~1024 workgroups need (4 per CU): every chunk is one more partial for the reduce pass
  const long by_grid = vr_cdiv(1024, (long)B * *ncb);
  if (nc > by_grid) nc = by_grid;
  if (nc > HW) nc = HW;
  *rows = vr_cdiv(HW, nc);
  *nchunks = (int)vr_cdiv(HW, *rows);
  return 0;
}

}  // namespace

extern "C" long vrnet_moments_workspace(int B, long HW, int C) {
  int TPR, ncb, nchunks;
  long rows;
  moments_plan(B, HW, C, 1, &TPR, &ncb, &nchunks, &rows);
  int nchunks4 = 0;
  if (C % 4 == 0) moments_plan(B, HW, C, 4, &TPR, &ncb, &nchunks4, &rows);      // the vector kernel may plan more chunks
  if (nchunks4 > nchunks) nchunks = nchunks4;
  return (long)B * nchunks * C * 2 * 8 + 256;
}

static int moments_launch(const float* x, long ldx, const float* x2, long ldx2, const float* mask, long ldm, int B,
                          long HW, int C, void* workspace, long workspace_bytes, hipStream_t st, int* nchunks_out,
                          int total_only = 0, const float* gamma = nullptr, double* gtot = nullptr, int* ncb_out = nullptr,
                          const float* mA = nullptr, const float* mD = nullptr, const float* mS = nullptr) {
  if (vr_ablated("moments")) { if (nchunks_out) *nchunks_out = 1; return VR_OK; }
  VR_CHECK_ARG(x && workspace, "moments: null tensor");
  VR_CHECK_ARG(B > 0 && HW > 0 && C > 0 && ldx >= C, "moments: bad shape");
  bool vec = (C % 4 == 0) && (ldx % 4 == 0) && vr_aligned16(x);
  if (x2) vec = vec && (ldx2 % 4 == 0) && vr_aligned16(x2);
  if (mask) vec = vec && (ldm % 4 == 0) && vr_aligned16(mask);
  int TPR, ncb, nchunks;
  long rows;
  moments_plan(B, HW, C, vec ? 4 : 1, &TPR, &ncb, &nchunks, &rows);
  if (workspace_bytes < vrnet_moments_workspace(B, HW, C)) {
    vr_set_error("moments: workspace too small");
    return VR_ERR_WORKSPACE;
  }
  double* partial = reinterpret_cast<double*>(workspace);
  dim3 grid(nchunks, B, ncb), block(256);
  if (vec)
    hipLaunchKernelGGL((moments_kernel<4>), grid, block, 256 * 8 * sizeof(double), st, x, ldx, x2, ldx2, mask, ldm, HW, C,
                       TPR, rows, nchunks, partial, total_only, gamma, gtot, mA, mD, mS);
  else
    hipLaunchKernelGGL((moments_kernel<1>), grid, block, 256 * 2 * sizeof(double), st, x, ldx, x2, ldx2, mask, ldm, HW, C,
                       TPR, rows, nchunks, partial, total_only, gamma, gtot, mA, mD, mS);
  VR_LAUNCH_CHECK("moments");
  if (ncb_out) *ncb_out = ncb;
  *nchunks_out = total_only ? nchunks * ncb : nchunks;
  return VR_OK;
}

extern "C" int vrnet_moments_f32(const float* x, long ldx, const float* x2, long ldx2, const float* mask, long ldm,
                                 int B, long HW, int C, double* out, void* workspace, long workspace_bytes,
                                 void* stream) {
  VR_CHECK_ARG(out, "moments: null tensor");
  hipStream_t st = vr_stream(stream);
  int nchunks;
  int rc = moments_launch(x, ldx, x2, ldx2, mask, ldm, B, HW, C, workspace, workspace_bytes, st, &nchunks);
  if (rc) return rc;
  const long n = (long)B * C * 2;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3(vr_cdiv(n, 16)), dim3(256), 0, st, reinterpret_cast<double*>(workspace), out,
                     B, nchunks, C);
  VR_LAUNCH_CHECK("moments_reduce");
  return VR_OK;
}

/* cat = shuffle_channels(torch.cat([ShuffleAttention(x), r], 1), 2) (vr_coc.py:343-349; the attention's gate from the P, Q, Mn
 * of vrnet_sa_coef_fwd) and mom[b][c] = (sum over the map of cat[b, :, c], 0) for the ECA gate behind it (:350): vrnet_sa_apply_f32 +
 * vrnet_cat2_f32 + vrnet_moments_f32 in two launches.  x, r: (B, HW, C) with row strides ldx, ldr; cat: (B, HW, 2 C), row stride
 * ldc; C % 4 == 0, C <= 512; workspace: vrnet_moments_workspace(B, HW, 2 C). */
extern "C" int vrnet_sa_cat_sums_f32(const float* x, long ldx, const float* P, const float* Q, const float* Mn, const float* r,
                                     long ldr, float* cat, long ldc, int B, long HW, int C, double* mom, void* workspace,
                                     long workspace_bytes, void* stream) {
  VR_CHECK_ARG(x && P && Q && Mn && r && cat && mom && workspace && B > 0 && HW > 0, "sa_cat_sums: null tensor");
  VR_CHECK_ARG(C % 4 == 0 && C >= 4 && C <= 512 && ldx >= C && ldr >= C && ldr % 2 == 0 && ldc >= 2 * C && ldc % 4 == 0 &&
                   vr_aligned16(cat) && (reinterpret_cast<uintptr_t>(r) & 7) == 0,
               "sa_cat_sums: needs C %% 4 == 0, C <= 512, an 8-byte aligned radar map and a 16-byte aligned output");
  if (workspace_bytes < vrnet_moments_workspace(B, HW, 2 * C)) {
    vr_set_error("sa_cat_sums: workspace too small");
    return VR_ERR_WORKSPACE;
  }
  if (vr_ablated("misc")) return VR_OK;
  int TPR, ncb, nchunks;
  long rows;
  moments_plan(B, HW, 2 * C, 4, &TPR, &ncb, &nchunks, &rows);       // (2 C / 4 quads per row = C / 2 threads: ncb == 1 up to C = 512)
  int tpr = 1;
  while (tpr < C / 2 && tpr < 256) tpr <<= 1;
  hipStream_t st = vr_stream(stream);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(sa_cat_sums_kernel, dim3(nchunks, B), dim3(256), 256 * 4 * sizeof(double), st, x, ldx, P, Q, Mn, r, ldr, cat, ldc,
                     HW, C, tpr, rows, nchunks, partial);
  VR_LAUNCH_CHECK("sa_cat_sums");
  const long n = (long)B * 2 * C * 2;
  hipLaunchKernelGGL(moments_reduce_kernel, dim3(vr_cdiv(n, 16)), dim3(256), 0, st, partial, mom, B, nchunks, 2 * C);
  VR_LAUNCH_CHECK("moments_reduce");
  return VR_OK;
}

extern "C" int vrnet_gn_coef_from_pairs(const double* pairs, long pairs_per_sample, const float* gamma, const float* beta,
                                        float eps, int B, long HW, int C, float* A, float* D, float* S, float* mean_rstd,
                                        const float* gamma2, const float* beta2, void* stream) {
  VR_CHECK_ARG(pairs && pairs_per_sample > 0 && gamma && beta && A && D && S && mean_rstd, "gn_coef_from_pairs: bad arguments");
  VR_CHECK_ARG((!gamma2 == !beta2) && (!gamma2 || B % 2 == 0), "gn_coef_from_pairs: two-stream launch needs gamma2, beta2, even batch");
  hipLaunchKernelGGL(gn_coef_fwd_partial_kernel, dim3(B), dim3(256), 0, vr_stream(stream), pairs, pairs_per_sample, gamma, beta,
                     eps, HW, C, A, D, S, mean_rstd, gamma2, beta2);
  VR_LAUNCH_CHECK("gn_coef_from_pairs");
  return VR_OK;
}

extern "C" int vrnet_gn_stats_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, int B,
                                  long HW, int C, float* A, float* D, float* S, float* mean_rstd, const float* gamma2,
                                  const float* beta2, void* workspace, long workspace_bytes, void* stream) {
  VR_CHECK_ARG(gamma && beta && A && D && S && mean_rstd, "gn_stats_fwd: null tensor");
  VR_CHECK_ARG((!gamma2 == !beta2) && (!gamma2 || B % 2 == 0), "gn_stats_fwd: two-stream launch needs gamma2, beta2, even batch");
  hipStream_t st = vr_stream(stream);
  int nchunks;
  int rc = moments_launch(x, ldx, nullptr, 0, nullptr, 0, B, HW, C, workspace, workspace_bytes, st, &nchunks, 1);
  if (rc) return rc;
  hipLaunchKernelGGL(gn_coef_fwd_partial_kernel, dim3(B), dim3(256), 0, st, reinterpret_cast<double*>(workspace),
                     (long)nchunks, gamma, beta, eps, HW, C, A, D, S, mean_rstd, gamma2, beta2);
  VR_LAUNCH_CHECK("gn_stats_fwd");
  return VR_OK;
}

static int affine_impl(const float* x1, long ld1, const float* A, const float* D1, const float* S1, int pre,
                       const float* masky, long ldm, const float* x2, long ld2, const float* E, const float* D2,
                       const float* S2, long coef_bstride, float* out, long ldo, int B, long HW, int C, int accumulate,
                       const float* add, long ldadd, const float* mA, const float* mD, const float* mS, void* stream) {
  VR_CHECK_ARG(out && B > 0 && HW > 0 && C > 0, "affine: bad arguments");
  if (vr_ablated("affine")) return VR_OK;
  VR_CHECK_ARG(pre != 2 || masky, "affine: mask mode without mask tensor");
  VR_CHECK_ARG(pre != 3 || (mA && mD && mS && x2), "affine: recomputed-mask mode needs z (x2) and the forward coefficients");
  VR_CHECK_ARG(!(accumulate && add), "affine: accumulate (in place) and add (out of place) are exclusive");
  if (accumulate) { add = out; ldadd = ldo; }
  AffineArgs p{x1, ld1, A, D1, S1, masky, ldm, x2, ld2, E, D2, S2, out, ldo, HW, C, coef_bstride, pre, accumulate, add, ldadd,
               mA, mD, mS};
  bool vec = (C % 4 == 0) && (ldo % 4 == 0) && vr_aligned16(out) && (coef_bstride % 4 == 0);
  if (x1) vec = vec && (ld1 % 4 == 0) && vr_aligned16(x1);
  if (x2) vec = vec && (ld2 % 4 == 0) && vr_aligned16(x2);
  if (pre == 2) vec = vec && (ldm % 4 == 0) && vr_aligned16(masky);
  if (add) vec = vec && (ldadd % 4 == 0) && vr_aligned16(add);
  for (const float* c : {A, D1, S1, E, D2, S2, mA, mD, mS}) vec = vec && (!c || vr_aligned16(c));
  // contiguous tensors of a channel count that is no multiple of 4: flat 16-byte accesses over a sample's HW * C floats
  bool flat = !vec && C % 4 != 0 && C <= 16 && (HW * C) % 4 == 0 && ldo == C && vr_aligned16(out) && (!x1 || (ld1 == C && vr_aligned16(x1))) &&
              (!x2 || (ld2 == C && vr_aligned16(x2))) && (pre != 2 || (ldm == C && vr_aligned16(masky))) &&
              (!add || (ldadd == C && vr_aligned16(add)));
  if (flat) {
    long blocks = vr_cdiv(HW * C / 4, 256 * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(affine_flat_kernel, dim3(blocks, B), dim3(256), 0, vr_stream(stream), p);
    VR_LAUNCH_CHECK("affine");
    return VR_OK;
  }
  long blocks = vr_cdiv(HW * (C / (vec ? 4 : 1)), 256 * 4);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  dim3 grid(blocks, B), block(256);
  if (vec) hipLaunchKernelGGL((affine_kernel<4>), grid, block, 0, vr_stream(stream), p);
  else hipLaunchKernelGGL((affine_kernel<1>), grid, block, 0, vr_stream(stream), p);
  VR_LAUNCH_CHECK("affine");
  return VR_OK;
}

extern "C" int vrnet_affine_f32(const float* x1, long ld1, const float* A, const float* D1, const float* S1, int pre,
                                const float* masky, long ldm, const float* x2, long ld2, const float* E,
                                const float* D2, const float* S2, long coef_bstride, float* out, long ldo, int B,
                                long HW, int C, int accumulate, const float* add, long ldadd, void* stream) {
  VR_CHECK_ARG(pre >= 0 && pre <= 2, "affine: pre 0 (none), 1 (ReLU) or 2 (mask by a ReLU output)");
  return affine_impl(x1, ld1, A, D1, S1, pre, masky, ldm, x2, ld2, E, D2, S2, coef_bstride, out, ldo, B, HW, C, accumulate, add,
                     ldadd, nullptr, nullptr, nullptr, stream);
}

extern "C" int vrnet_gn_coef_fwd(const double* mom, const float* gamma, const float* beta, float eps, int B, long HW,
                                 int C, float* A, float* D, float* S, float* mean_rstd, void* stream) {
  VR_CHECK_ARG(mom && gamma && beta && A && D && S && mean_rstd, "gn_coef_fwd: null tensor");
  hipLaunchKernelGGL(gn_coef_fwd_kernel, dim3(B), dim3(256), 0, vr_stream(stream), mom, gamma, beta, eps, HW, C, A, D, S,
                     mean_rstd, (const float*)nullptr, (const float*)nullptr);
  VR_LAUNCH_CHECK("gn_coef_fwd");
  return VR_OK;
}

/* GroupNorm(1, C) forward from the producer's tile statistics, ONE launch: y = GN(x); mean_rstd [B][2] for the backward
 * pass.  pairs: `pairs_per_sample` consecutive fp64 (sum, sumsq) pairs per sample (vrnet_conv2d_f32 / vrnet_mlp_fwd_f32
 * `stats`).  Needs C % 4 == 0 and 16-byte rows. */
static int gn_apply_fwd_impl(const float* x, long ldx, const double* pairs, long pairs_per_sample, const float* gamma,
                             const float* beta, float eps, int B, long HW, int C, float* y, long ldy, float* mean_rstd,
                             const vrnet_planes_out* yp, void* stream) {
  VR_CHECK_ARG(x && pairs && gamma && beta && (y || yp) && mean_rstd, "gn_apply_fwd: null tensor");
  VR_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C && (!y || (ldy % 4 == 0 && ldy >= C && vr_aligned16(y))) &&
                   vr_aligned16(x) && vr_aligned16(gamma) && vr_aligned16(beta) && vr_planes_out_ok(yp, C),
               "gn_apply_fwd: needs C %% 4 == 0 and 16-byte aligned rows");
  if (vr_ablated("affine") || vr_ablated("gnfwd")) return VR_OK;
  long bx = vr_cdiv(HW * (C / 4), 256 * 4);            // ~4 float4 per thread: the pair reduction is repeated per workgroup
  const long cap = vr_cdiv(2048, B);
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  const vrnet_planes_out none{};
  hipLaunchKernelGGL(gn_apply_fwd_kernel, dim3((unsigned)bx, B), dim3(256), 0, vr_stream(stream), x, ldx, pairs, pairs_per_sample,
                     gamma, beta, eps, HW, C, y, ldy, mean_rstd, yp ? *yp : none);
  VR_LAUNCH_CHECK("gn_apply_fwd");
  return VR_OK;
}
extern "C" int vrnet_gn_apply_fwd(const float* x, long ldx, const double* pairs, long pairs_per_sample, const float* gamma,
                                  const float* beta, float eps, int B, long HW, int C, float* y, long ldy, float* mean_rstd,
                                  void* stream) {
  VR_CHECK_ARG(y, "gn_apply_fwd: null tensor");
  return gn_apply_fwd_impl(x, ldx, pairs, pairs_per_sample, gamma, beta, eps, B, HW, C, y, ldy, mean_rstd, nullptr, stream);
}
/* The same with the result (also, or only: y may be NULL) written as bf16 planes -- the operand format of the plane GEMMs
 * (vrnet_gemm_planes_f32 / vrnet_wgrad_planes_f32): the conv behind the GroupNorm then splits nothing. */
extern "C" int vrnet_gn_apply_fwd_planes(const float* x, long ldx, const double* pairs, long pairs_per_sample, const float* gamma,
                                         const float* beta, float eps, int B, long HW, int C, float* y, long ldy,
                                         float* mean_rstd, const vrnet_planes_out* yp, void* stream) {
  return gn_apply_fwd_impl(x, ldx, pairs, pairs_per_sample, gamma, beta, eps, B, HW, C, y, ldy, mean_rstd, yp, stream);
}

/* GroupNorm(1, C) backward in TWO launches (moments of (dy, dy * x) with gamma-weighted chunk totals; apply + parameter
 * gradients): out = dx (+ add), dgamma / dbeta (accumulated when accumulate_params).  workspace: vrnet_gn_bwd_workspace. */
extern "C" long vrnet_gn_bwd_workspace(int B, long HW, int C) {
  return vrnet_moments_workspace(B, HW, C) + (long)B * 512 * 4 * 2 * 8 + 256;
}
static int gn_apply_bwd_impl(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd,
                             const float* gamma, int B, long HW, int C, const float* add, long ldadd, float* out, long ldo,
                             float* dgamma, float* dbeta, int accumulate_params, const vrnet_planes_out* outp, void* workspace,
                             long workspace_bytes, void* stream);
extern "C" int vrnet_gn_apply_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd,
                                  const float* gamma, int B, long HW, int C, const float* add, long ldadd, float* out, long ldo,
                                  float* dgamma, float* dbeta, int accumulate_params, void* workspace, long workspace_bytes,
                                  void* stream) {
  return gn_apply_bwd_impl(dy, lddy, x, ldx, mean_rstd, gamma, B, HW, C, add, ldadd, out, ldo, dgamma, dbeta, accumulate_params,
                           nullptr, workspace, workspace_bytes, stream);
}
/* The same with a second copy of `out` as bf16 planes: the gradient a block hands to the block before it is the dy operand
 * of that block's data- and weight-gradient GEMMs. */
extern "C" int vrnet_gn_apply_bwd_planes(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd,
                                         const float* gamma, int B, long HW, int C, const float* add, long ldadd, float* out,
                                         long ldo, float* dgamma, float* dbeta, int accumulate_params,
                                         const vrnet_planes_out* outp, void* workspace, long workspace_bytes, void* stream) {
  return gn_apply_bwd_impl(dy, lddy, x, ldx, mean_rstd, gamma, B, HW, C, add, ldadd, out, ldo, dgamma, dbeta, accumulate_params,
                           outp, workspace, workspace_bytes, stream);
}
static int gn_apply_bwd_impl(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd,
                             const float* gamma, int B, long HW, int C, const float* add, long ldadd, float* out, long ldo,
                             float* dgamma, float* dbeta, int accumulate_params, const vrnet_planes_out* outp, void* workspace,
                             long workspace_bytes, void* stream) {
  if (vr_ablated("affine") || vr_ablated("gnbwd")) return VR_OK;
  VR_CHECK_ARG(dy && x && mean_rstd && gamma && out && dgamma && dbeta && workspace && vr_planes_out_ok(outp, C), "gn_apply_bwd: null tensor");
  VR_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && (!add || ldadd % 4 == 0) &&
                   vr_aligned16(dy) && vr_aligned16(x) && vr_aligned16(out) && vr_aligned16(gamma) && (!add || vr_aligned16(add)),
               "gn_apply_bwd: needs C %% 4 == 0 and 16-byte aligned rows");
  if (workspace_bytes < vrnet_gn_bwd_workspace(B, HW, C)) {
    vr_set_error("gn_apply_bwd: workspace too small");
    return VR_ERR_WORKSPACE;
  }
  hipStream_t st = vr_stream(stream);
  const long mom_bytes = vrnet_moments_workspace(B, HW, C);
  double* gtot = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + ((mom_bytes + 255) / 256) * 256);
  int nchunks = 1, ncb = 1;
  int rc = moments_launch(dy, lddy, x, ldx, nullptr, 0, B, HW, C, workspace, mom_bytes, st, &nchunks, 0, gamma, gtot, &ncb);
  if (rc) return rc;
  if (vr_ablated("affine")) return VR_OK;
  long bx = vr_cdiv(HW * (C / 4), 256 * 4);
  const long cap = vr_cdiv(2048, B);
  if (bx > cap) bx = cap;
  const long need = vr_cdiv(C, 4);                      // the parameter-gradient row of the grid: one wave per channel
  if (bx < need) bx = need;
  hipLaunchKernelGGL(gn_apply_bwd_kernel, dim3((unsigned)bx, B + 1), dim3(256), 0, st, dy, lddy, x, ldx, gtot, nchunks * ncb,
                     reinterpret_cast<double*>(workspace), nchunks, mean_rstd, gamma, B, HW, C, add, ldadd, out, ldo, dgamma, dbeta,
                     accumulate_params, outp ? *outp : vrnet_planes_out{});
  VR_LAUNCH_CHECK("gn_apply_bwd");
  return VR_OK;
}

/* The apply step alone, for dy whose moments the producing conv already left (vrnet_conv2d_f32 colstats with x2 = x and
 * gamma): partial [B * HW/32][C][2], tile_totals [B * HW/32][ceil(C/32)][2]; HW % 32 == 0.  ONE launch. */
extern "C" int vrnet_gn_apply_bwd_from_partials(const float* dy, long lddy, const float* x, long ldx, const double* partial,
                                                const double* tile_totals, const float* mean_rstd, const float* gamma, int B,
                                                long HW, int C, const float* add, long ldadd, float* out, long ldo,
                                                float* dgamma, float* dbeta, int accumulate_params, void* stream) {
  VR_CHECK_ARG(dy && x && partial && tile_totals && mean_rstd && gamma && out && dgamma && dbeta, "gn_apply_bwd_from_partials: null tensor");
  VR_CHECK_ARG(B > 0 && HW > 0 && HW % 32 == 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 &&
                   (!add || ldadd % 4 == 0) && vr_aligned16(dy) && vr_aligned16(x) && vr_aligned16(out) && vr_aligned16(gamma) &&
                   (!add || vr_aligned16(add)),
               "gn_apply_bwd_from_partials: needs HW %% 32 == 0, C %% 4 == 0 and 16-byte aligned rows");
  if (vr_ablated("affine")) return VR_OK;
  long bx = vr_cdiv(HW * (C / 4), 256 * 4);
  const long cap = vr_cdiv(2048, B);
  if (bx > cap) bx = cap;
  const long need = vr_cdiv(C, 4);
  if (bx < need) bx = need;
  const int nchunks = (int)(HW / 32);
  hipLaunchKernelGGL(gn_apply_bwd_kernel, dim3((unsigned)bx, B + 1), dim3(256), 0, vr_stream(stream), dy, lddy, x, ldx, tile_totals,
                     nchunks * (int)vr_cdiv(C, 32), partial, nchunks, mean_rstd, gamma, B, HW, C, add, ldadd, out, ldo, dgamma, dbeta,
                     accumulate_params, vrnet_planes_out{});
  VR_LAUNCH_CHECK("gn_apply_bwd_from_partials");
  return VR_OK;
}

/* Train-mode BatchNorm coefficients + running statistics from the column partials the producing conv left
 * (vrnet_conv2d_f32 colstats, x2 = NULL): partial [B * HW/32][C][2]; HW % 32 == 0.  ONE launch, no pass over the tensor. */
extern "C" int vrnet_bn_coef_fwd_from_partials(const double* partial, const float* gamma, const float* beta, float eps,
                                               float momentum, float* running_mean, float* running_var,
                                               long long* num_batches_tracked, int B, long HW, int C, float* A, float* D, float* S,
                                               float* mean_rstd, void* stream) {
  VR_CHECK_ARG(partial && gamma && beta && running_mean && running_var && A && D && S && mean_rstd, "bn_coef_fwd_from_partials: null tensor");
  VR_CHECK_ARG((long)B * HW > 1 && HW % 32 == 0, "bn_coef_fwd_from_partials: needs HW %% 32 == 0 and more than 1 value per channel");
  hipLaunchKernelGGL(bn_coef_fwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, vr_stream(stream), partial, (int)(HW / 32), gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches_tracked, B, HW, C, A, D, S, mean_rstd);
  VR_LAUNCH_CHECK("bn_coef_fwd_from_partials");
  return VR_OK;
}

/* The same two coefficient steps from column partials with an explicit chunk count (round 5: the fused kernels of
 * csrc/fusion.hip leave [nchunks][C][2] partials over ALL rows of the batch): train-mode BatchNorm forward coefficients +
 * running statistics (count = B * HW values per channel), and the backward coefficients + parameter gradients from
 * (sum dy', sum dy' z) partials. */
extern "C" int vrnet_bn_coef_fwd_from_chunks(const double* partial, int nchunks, long count, const float* gamma, const float* beta,
                                             float eps, float momentum, float* running_mean, float* running_var,
                                             long long* num_batches_tracked, int C, float* A, float* D, float* S, float* mean_rstd,
                                             void* stream) {
  VR_CHECK_ARG(partial && nchunks > 0 && gamma && beta && running_mean && running_var && A && D && S && mean_rstd && count > 1 && C > 0,
               "bn_coef_fwd_from_chunks: bad arguments");
  hipLaunchKernelGGL(bn_coef_fwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, vr_stream(stream), partial, nchunks, gamma, beta,
                     eps, momentum, running_mean, running_var, num_batches_tracked, 1, count, C, A, D, S, mean_rstd);
  VR_LAUNCH_CHECK("bn_coef_fwd_from_chunks");
  return VR_OK;
}
extern "C" int vrnet_bn_coef_bwd_from_chunks(const double* partial, int nchunks, long count, const float* mean_rstd,
                                             const float* gamma, int training, int C, float* A, float* E, float* D, float* S,
                                             float* dgamma, float* dbeta, int accumulate, void* stream) {
  VR_CHECK_ARG(partial && nchunks > 0 && mean_rstd && gamma && A && E && D && S && dgamma && dbeta && count > 0 && C > 0,
               "bn_coef_bwd_from_chunks: bad arguments");
  hipLaunchKernelGGL(bn_coef_bwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, vr_stream(stream), partial, nchunks, mean_rstd,
                     gamma, training, 1, count, C, A, E, D, S, dgamma, dbeta, accumulate);
  VR_LAUNCH_CHECK("bn_coef_bwd_from_chunks");
  return VR_OK;
}

extern "C" int vrnet_gn_coef_bwd(const double* mom2, const float* mean_rstd, const float* gamma, int B, long HW, int C,
                                 float* A, float* E, float* D, float* S, float* dgamma, float* dbeta, int accumulate,
                                 const float* gamma2, float* dgamma2, float* dbeta2, void* stream) {
  VR_CHECK_ARG(mom2 && mean_rstd && gamma && A && E && D && S && dgamma && dbeta, "gn_coef_bwd: null tensor");
  VR_CHECK_ARG((!gamma2 == !dgamma2) && (!gamma2 == !dbeta2) && (!gamma2 || B % 2 == 0),
               "gn_coef_bwd: two-stream launch needs gamma2, dgamma2, dbeta2 and an even batch");
  hipLaunchKernelGGL(gn_coef_bwd_kernel, dim3(B + vr_cdiv(C, 256)), dim3(256), 0, vr_stream(stream), mom2, mean_rstd,
                     gamma, B, HW, C, A, E, D, S, dgamma, dbeta, accumulate, gamma2, dgamma2, dbeta2);
  VR_LAUNCH_CHECK("gn_coef_bwd");
  return VR_OK;
}

extern "C" int vrnet_bn_coef_fwd(const double* mom, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, long long* num_batches_tracked,
                                 int training, int B, long HW, int C, float* A, float* D, float* S, float* mean_rstd,
                                 void* stream) {
  VR_CHECK_ARG(gamma && beta && running_mean && running_var && A && D && S && mean_rstd, "bn_coef_fwd: null tensor");
  VR_CHECK_ARG(!training || mom, "bn_coef_fwd: training mode needs batch moments");
  VR_CHECK_ARG(!training || (long)B * HW > 1, "Expected more than 1 value per channel when training");
  hipLaunchKernelGGL(bn_coef_fwd_kernel, dim3(vr_cdiv(C, 128)), dim3(128), 0, vr_stream(stream), mom, gamma, beta, eps,
                     momentum, running_mean, running_var, num_batches_tracked, training, B, HW, C, A, D, S, mean_rstd);
  VR_LAUNCH_CHECK("bn_coef_fwd");
  return VR_OK;
}

// Train-mode BatchNorm statistics + coefficients in two launches (moments over x, then reduce + coefficients + running
// statistics); workspace as vrnet_moments_workspace.
extern "C" int vrnet_bn_stats_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, int B, long HW,
                                  int C, float* A, float* D, float* S, float* mean_rstd, void* workspace,
                                  long workspace_bytes, void* stream) {
  VR_CHECK_ARG(gamma && beta && running_mean && running_var && A && D && S && mean_rstd, "bn_stats_fwd: null tensor");
  VR_CHECK_ARG((long)B * HW > 1, "Expected more than 1 value per channel when training");
  hipStream_t st = vr_stream(stream);
  int nchunks;
  int rc = moments_launch(x, ldx, nullptr, 0, nullptr, 0, B, HW, C, workspace, workspace_bytes, st, &nchunks);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_coef_fwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, st, reinterpret_cast<double*>(workspace),
                     nchunks, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, B, HW, C, A, D, S,
                     mean_rstd);
  VR_LAUNCH_CHECK("bn_stats_fwd");
  return VR_OK;
}

// Backward counterpart: moments of (dy [masked by the ReLU output], dy * z), then reduce + coefficients + d gamma / d beta.
extern "C" int vrnet_bn_stats_bwd(const float* dy, long lddy, const float* z, long ldz, const float* mask, long ldm,
                                  const float* mean_rstd, const float* gamma, int training, int B, long HW, int C, float* A,
                                  float* E, float* D, float* S, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                  long workspace_bytes, void* stream) {
  VR_CHECK_ARG(dy && z && mean_rstd && gamma && A && E && D && S && dgamma && dbeta, "bn_stats_bwd: null tensor");
  hipStream_t st = vr_stream(stream);
  int nchunks;
  int rc = moments_launch(dy, lddy, z, ldz, mask, ldm, B, HW, C, workspace, workspace_bytes, st, &nchunks);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_coef_bwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, st, reinterpret_cast<double*>(workspace),
                     nchunks, mean_rstd, gamma, training, B, HW, C, A, E, D, S, dgamma, dbeta, accumulate);
  VR_LAUNCH_CHECK("bn_stats_bwd");
  return VR_OK;
}

/* BatchNorm backward of y = ReLU(BN(z)) WITHOUT reading y (round 4): the mask [y > 0] is recomputed from z with the forward
 * coefficients fwd_A (z - fwd_S) + fwd_D -- the single fused multiply-add the forward apply evaluated, so the bits are the
 * forward's -- which takes one of the three tensor reads out of the moments pass and one of three out of the apply pass.
 * vrnet_bn_stats_bwd_zmask = vrnet_bn_stats_bwd; vrnet_bn_apply_bwd_zmask: dz = [mask] (A dy) + E (z - S) + D. */
extern "C" int vrnet_bn_stats_bwd_zmask(const float* dy, long lddy, const float* z, long ldz, const float* fwd_A,
                                        const float* fwd_D, const float* fwd_S, const float* mean_rstd, const float* gamma,
                                        int training, int B, long HW, int C, float* A, float* E, float* D, float* S,
                                        float* dgamma, float* dbeta, int accumulate, void* workspace, long workspace_bytes,
                                        void* stream) {
  VR_CHECK_ARG(dy && z && fwd_A && fwd_D && fwd_S && mean_rstd && gamma && A && E && D && S && dgamma && dbeta,
               "bn_stats_bwd_zmask: null tensor");
  VR_CHECK_ARG(C % 4 != 0 || (vr_aligned16(fwd_A) && vr_aligned16(fwd_D) && vr_aligned16(fwd_S)),
               "bn_stats_bwd_zmask: coefficient vectors must be 16-byte aligned");
  hipStream_t st = vr_stream(stream);
  int nchunks;
  int rc = moments_launch(dy, lddy, z, ldz, nullptr, 0, B, HW, C, workspace, workspace_bytes, st, &nchunks, 0, nullptr, nullptr,
                          nullptr, fwd_A, fwd_D, fwd_S);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_coef_bwd_partial_kernel, dim3(vr_cdiv(C, 4)), dim3(256), 0, st, reinterpret_cast<double*>(workspace),
                     nchunks, mean_rstd, gamma, training, B, HW, C, A, E, D, S, dgamma, dbeta, accumulate);
  VR_LAUNCH_CHECK("bn_stats_bwd_zmask");
  return VR_OK;
}

extern "C" int vrnet_bn_apply_bwd_zmask(const float* dy, long lddy, const float* z, long ldz, const float* fwd_A,
                                        const float* fwd_D, const float* fwd_S, const float* A, const float* E,
                                        const float* D, const float* S, float* dz, long lddz, int B, long HW, int C,
                                        void* stream) {
  VR_CHECK_ARG(dy && z && fwd_A && fwd_D && fwd_S && A && E && D && S && dz, "bn_apply_bwd_zmask: null tensor");
  return affine_impl(dy, lddy, A, nullptr, nullptr, 3, nullptr, 0, z, ldz, E, D, S, 0, dz, lddz, B, HW, C, 0, nullptr, 0, fwd_A,
                     fwd_D, fwd_S, stream);
}

extern "C" int vrnet_bn_coef_bwd(const double* mom2, const float* mean_rstd, const float* gamma, int training, int B,
                                 long HW, int C, float* A, float* E, float* D, float* S, float* dgamma, float* dbeta,
                                 int accumulate, void* stream) {
  VR_CHECK_ARG(mom2 && mean_rstd && gamma && A && E && D && S && dgamma && dbeta, "bn_coef_bwd: null tensor");
  hipLaunchKernelGGL(bn_coef_bwd_kernel, dim3(vr_cdiv(C, 128)), dim3(128), 0, vr_stream(stream), mom2, mean_rstd, gamma,
                     training, B, HW, C, A, E, D, S, dgamma, dbeta, accumulate);
  VR_LAUNCH_CHECK("bn_coef_bwd");
  return VR_OK;
}

extern "C" int vrnet_eca_coef_fwd(const double* mom, const float* wk, int k, int B, long HW, int C, float* gate,
                                  void* stream) {
  VR_CHECK_ARG(mom && wk && gate && k >= 1 && (k & 1), "eca_coef_fwd: bad arguments");
  hipLaunchKernelGGL(eca_coef_fwd_kernel, dim3(vr_cdiv((long)B * C, 256)), dim3(256), 0, vr_stream(stream), mom, wk, k,
                     B, HW, C, gate);
  VR_LAUNCH_CHECK("eca_coef_fwd");
  return VR_OK;
}

extern "C" int vrnet_eca_coef_bwd(const double* mom2, const double* mom, const float* gate, const float* wk, int k,
                                  int B, long HW, int C, float* F, float* dwk, int accumulate, void* stream) {
  // F == NULL or dwk == NULL (ABI 9): only the other half -- the kernel-weight gradient is needed by nobody on the backward
  // chain, so the caller may issue it on a side stream
  VR_CHECK_ARG(mom2 && mom && gate && wk && (F || dwk), "eca_coef_bwd: null tensor");
  if (F) {
    hipLaunchKernelGGL(eca_coef_bwd_kernel, dim3(vr_cdiv((long)B * C, 256)), dim3(256), 0, vr_stream(stream), mom2, gate,
                       wk, k, B, HW, C, F);
    VR_LAUNCH_CHECK("eca_coef_bwd");
  }
  if (dwk) {
    hipLaunchKernelGGL(eca_dwk_kernel, dim3(k), dim3(256), 0, vr_stream(stream), mom2, mom, gate, k, B, HW, C, dwk,
                       accumulate);
    VR_LAUNCH_CHECK("eca_dwk");
  }
  return VR_OK;
}

extern "C" int vrnet_ls_coef_bwd(const double* mom2, const float* ls, int B, int C, float* dls, float* dbias,
                                 int accumulate, int pair, const float* ls2, float* dls2, float* dbias2, void* stream) {
  VR_CHECK_ARG(mom2, "ls_coef_bwd: null tensor");
  VR_CHECK_ARG(!pair || (B % 2 == 0 && (!ls == !ls2) && (!dls == !dls2) && (!dbias == !dbias2)),
               "ls_coef_bwd: two-stream launch needs the second parameter set and an even batch");
  hipLaunchKernelGGL(ls_coef_bwd_kernel, dim3(vr_cdiv(C, 128), pair ? 2 : 1), dim3(128), 0, vr_stream(stream), mom2, ls, B, C,
                     dls, dbias, accumulate, ls2, dls2, dbias2);
  VR_LAUNCH_CHECK("ls_coef_bwd");
  return VR_OK;
}

extern "C" int vrnet_moments_to_float(const double* mom, float* out, long n, double scale, int which, void* stream) {
  VR_CHECK_ARG(mom && out && (which == 0 || which == 1), "moments_to_float: bad arguments");
  hipLaunchKernelGGL(moments_to_float_kernel, dim3(vr_cdiv(n, 256)), dim3(256), 0, vr_stream(stream), mom, out, n, scale,
                     which);
  VR_LAUNCH_CHECK("moments_to_float");
  return VR_OK;
}

extern "C" int vrnet_copy_channels_f32(const float* src, long lds, int scs, float* dst, long ldd, int dcs, long rows,
                                       int C, int accumulate, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(src && dst && rows > 0 && C > 0, "copy_channels: bad arguments");
  long blocks = vr_cdiv(rows * C, 1024);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(blocks), dim3(256), 0, vr_stream(stream), src, lds, scs, dst, ldd, dcs,
                     rows, C, accumulate);
  VR_LAUNCH_CHECK("copy_channels");
  return VR_OK;
}

extern "C" int vrnet_cat2_f32(float* a, long lda, int Ca, float* b, long ldb, int Cb, float* cat, long ldc, long rows,
                              int interleave, int dir, int accumulate_a, int accumulate_b, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(cat && rows > 0 && Ca > 0 && Cb > 0 && (a || b) && (dir == 0 || dir == 1), "cat2: bad arguments");
  VR_CHECK_ARG(dir == 1 || (a && b), "cat2: the forward direction needs both sources");
  VR_CHECK_ARG(!interleave || Ca == Cb, "cat2: the channel shuffle needs halves of equal width");
  VR_CHECK_ARG((!a || lda >= Ca) && (!b || ldb >= Cb) && ldc >= Ca + Cb, "cat2: row strides");
  const bool vec = Ca % 4 == 0 && Cb % 4 == 0 && ldc % 4 == 0 && vr_aligned16(cat) && (!a || (lda % 4 == 0 && vr_aligned16(a))) &&
                   (!b || (ldb % 4 == 0 && vr_aligned16(b)));
  if (vec) {
    long vblocks = vr_cdiv(rows * (interleave ? Ca / 4 : (Ca + Cb) / 4), 256 * 2);
    if (vblocks > 8192) vblocks = 8192;
    if (vblocks < 1) vblocks = 1;
    hipLaunchKernelGGL(cat2_vec_kernel, dim3(vblocks), dim3(256), 0, vr_stream(stream), a, lda, Ca, b, ldb, Cb, cat, ldc, rows,
                       interleave, dir, accumulate_a, accumulate_b);
    VR_LAUNCH_CHECK("cat2");
    return VR_OK;
  }
  if (Ca == 3 && Cb == 4 && !interleave) {      // the input fusion's concat: one thread per row
    long rb = vr_cdiv(rows, 256);
    if (rb > 16384) rb = 16384;
    const int vecb = (!b || (ldb % 4 == 0 && vr_aligned16(b))) ? 1 : 0;
    hipLaunchKernelGGL((cat2_rows_kernel<3, 4>), dim3(rb), dim3(256), 0, vr_stream(stream), a, lda, b, ldb, cat, ldc, rows, dir,
                       accumulate_a, accumulate_b, vecb);
    VR_LAUNCH_CHECK("cat2");
    return VR_OK;
  }
  long blocks = vr_cdiv(rows * (Ca + Cb), 1024);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cat2_kernel, dim3(blocks), dim3(256), 0, vr_stream(stream), a, lda, Ca, b, ldb, Cb, cat, ldc, rows,
                     interleave, dir, accumulate_a, accumulate_b);
  VR_LAUNCH_CHECK("cat2");
  return VR_OK;
}

extern "C" int vrnet_nchw_to_nhwc_f32(const float* src, float* dst, long ldd, int B, int C, long HW, void* stream) {
  VR_CHECK_ARG(src && dst && ldd >= C, "nchw_to_nhwc: bad arguments");
  if ((C == 3 || C == 4) && HW >= 4096) {
    long pb = vr_cdiv(HW, 256);
    if (pb > 4096) pb = 4096;
    if (C == 3) hipLaunchKernelGGL((nchw_to_nhwc_small_kernel<3>), dim3(pb, B), dim3(256), 0, vr_stream(stream), src, dst, ldd, HW, 0);
    else hipLaunchKernelGGL((nchw_to_nhwc_small_kernel<4>), dim3(pb, B), dim3(256), 0, vr_stream(stream), src, dst, ldd, HW,
                            (ldd % 4 == 0 && vr_aligned16(dst)) ? 1 : 0);
    VR_LAUNCH_CHECK("nchw_to_nhwc");
    return VR_OK;
  }
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(vr_cdiv(HW, 32), vr_cdiv(C, 32), B), dim3(32, 8), 0, vr_stream(stream),
                     src, dst, ldd, C, HW);
  VR_LAUNCH_CHECK("nchw_to_nhwc");
  return VR_OK;
}

extern "C" int vrnet_nhwc_to_nchw_f32(const float* src, long lds, float* dst, int B, int C, long HW, int accumulate,
                                      void* stream) {
  VR_CHECK_ARG(src && dst && lds >= C, "nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(vr_cdiv(HW, 32), vr_cdiv(C, 32), B), dim3(32, 8), 0, vr_stream(stream),
                     src, lds, dst, C, HW, accumulate);
  VR_LAUNCH_CHECK("nhwc_to_nchw");
  return VR_OK;
}

extern "C" int vrnet_add_f32(float* dst, const float* src, long n, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(dst && src && n >= 0, "add: bad arguments");
  if (n == 0) return VR_OK;
  long blocks = vr_cdiv(n, 1024);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, vr_stream(stream), dst, src, n);
  VR_LAUNCH_CHECK("add");
  return VR_OK;
}

extern "C" int vrnet_fill_f32(float* dst, float value, long n, void* stream) {
  VR_CHECK_ARG(dst && n >= 0, "fill: bad arguments");
  if (n == 0) return VR_OK;
  long blocks = vr_cdiv(n, 1024);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, vr_stream(stream), dst, value, n);
  VR_LAUNCH_CHECK("fill");
  return VR_OK;
}
