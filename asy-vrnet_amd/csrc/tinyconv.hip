// Direct (no MFMA) kernels for the tiny-channel layers at full input resolution, where an MFMA tile would be
// > 99 % padding and the op is a pure HBM stream: image_initial / radar_initial 1x1 (3->3, 4->4),
// radar_projection 3x3 4->3, inverse_projection 1x1 7->4 (backbone/fusion/vr_coc.py:415-422,308,327) and the
// data gradient of the 4x4/s4 patch embeddings (5->64, 6->64; vr_coc.py:424-430).  One thread per pixel,
// weights in LDS, all channels of the pixel in registers.
#include "common.h"

namespace tiny {

constexpr int CMAX = 8;

struct Args {
  const float* a; long lda; const float* w; const float* bias; float* y; long ldy;
  int B, H, W, Cin, Cout, k, pad, dil, accumulate;
};

// MODE 0: y[pix][n] = bias[n] + sum_{t,c} x[pix + off_t][c] w[t][n][c]        (stride 1, same size)
// MODE 1: dx[pix][c] = sum_{t,n} dy[pix - off_t][n] w[t][n][c]
template <int MODE>
__global__ __launch_bounds__(256) void conv_kernel(const Args p) {
  __shared__ float ws[9 * CMAX * CMAX];
  const int T = p.k * p.k;
  for (int i = threadIdx.x; i < T * p.Cout * p.Cin; i += 256) ws[i] = p.w[i];
  __syncthreads();
  const int CK = MODE == 0 ? p.Cin : p.Cout, CN = MODE == 0 ? p.Cout : p.Cin;
  const long total = (long)p.B * p.H * p.W;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int x = pix % p.W;
    const long q = pix / p.W;
    const int y = q % p.H;
    const long b = q / p.H;
    float acc[CMAX];
#pragma unroll
    for (int n = 0; n < CMAX; ++n) acc[n] = (MODE == 0 && p.bias && n < CN) ? p.bias[n] : 0.f;
    for (int ky = 0; ky < p.k; ++ky) {
      const int oy = ky * p.dil - p.pad;
      const int sy = MODE == 0 ? y + oy : y - oy;
      if (sy < 0 || sy >= p.H) continue;
      for (int kx = 0; kx < p.k; ++kx) {
        const int ox = kx * p.dil - p.pad;
        const int sx = MODE == 0 ? x + ox : x - ox;
        if (sx < 0 || sx >= p.W) continue;
        const float* src = p.a + ((b * p.H + sy) * p.W + sx) * p.lda;
        const float* wt = ws + (ky * p.k + kx) * p.Cout * p.Cin;
        float v[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) v[c] = c < CK ? src[c] : 0.f;
#pragma unroll
        for (int n = 0; n < CMAX; ++n) {
          if (n >= CN) break;
#pragma unroll
          for (int c = 0; c < CMAX; ++c) {
            if (c >= CK) break;
            acc[n] += v[c] * (MODE == 0 ? wt[n * p.Cin + c] : wt[c * p.Cin + n]);
          }
        }
      }
    }
    float* d = p.y + pix * p.ldy;
#pragma unroll
    for (int n = 0; n < CMAX; ++n)
      if (n < CN) d[n] = p.accumulate ? d[n] + acc[n] : acc[n];
  }
}

// The four layer shapes of the input fusion with every loop bound a compile-time constant (round 5): image_initial 3 -> 3 and
// radar_initial 4 -> 4 (1 x 1), radar_projection 4 -> 3 (3 x 3), inverse_projection 7 -> 4 (1 x 1), forward and data gradient.
// CK = contracted channels, CN = produced channels.  The generic kernel above keeps every channel loop at CMAX with runtime
// breaks (and ran these 2 M-pixel launches -- all of them on the step's critical chain -- at 1-1.4 TB/s); here the taps and
// channels unroll completely, 4-channel rows move as one 16-byte access, and the weights sit in LDS as broadcast reads.
template <int MODE, int CK, int CN, int KS>
__global__ __launch_bounds__(256) void conv_fixed_kernel(const Args p) {
  constexpr int T = KS * KS;
  constexpr int CO = MODE == 0 ? CN : CK, CI = MODE == 0 ? CK : CN;      // the layer's (Cout, Cin): weights are [t][Cout][Cin]
  __shared__ float ws[T * CO * CI];
  for (int i = threadIdx.x; i < T * CO * CI; i += 256) ws[i] = p.w[i];
  __syncthreads();
  const int total = p.B * p.H * p.W;                 // (< 2^31: checked by the caller)
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < total; pix += gridDim.x * 256) {
    const int x = pix % p.W;
    const int q = pix / p.W;
    const int y = q % p.H;
    float acc[CN];
#pragma unroll
    for (int n = 0; n < CN; ++n) acc[n] = (MODE == 0 && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int oy = ky * p.dil - p.pad;
      const int sy = MODE == 0 ? y + oy : y - oy;
      if (KS > 1 && (sy < 0 || sy >= p.H)) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int ox = kx * p.dil - p.pad;
        const int sx = MODE == 0 ? x + ox : x - ox;
        if (KS > 1 && (sx < 0 || sx >= p.W)) continue;
        const float* src = p.a + ((long)pix + (long)(sy - y) * p.W + (sx - x)) * p.lda;
        const float* wt = ws + (ky * KS + kx) * CO * CI;
        float v[CK];
        if (CK == 4 && p.lda == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int c = 0; c < CK; ++c) v[c] = t[c];
        } else {
#pragma unroll
          for (int c = 0; c < CK; ++c) v[c] = src[c];
        }
#pragma unroll
        for (int n = 0; n < CN; ++n)
#pragma unroll
          for (int c = 0; c < CK; ++c) acc[n] += v[c] * (MODE == 0 ? wt[n * CI + c] : wt[c * CI + n]);
      }
    }
    float* d = p.y + (long)pix * p.ldy;
    if (CN == 4 && p.ldy == 4) {
      f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
      if (p.accumulate) {
        const f32x4 old = *reinterpret_cast<const f32x4*>(d);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] += old[n];
      }
      *reinterpret_cast<f32x4*>(d) = o;
    } else {
#pragma unroll
      for (int n = 0; n < CN; ++n) d[n] = p.accumulate ? d[n] + acc[n] : acc[n];
    }
  }
}

// partial[blk][n][Cin*T + 1]: sum over this block's pixels of dy[pix][n] * x[pix + off_t][c], last = sum dy (bias).
// CP = channel count padded to 4 / 8, KS = 1 / 3 (compile-time: the accumulators stay in registers and the tap
// loops unroll); VEC4: Cin == 4 on 16-byte rows -> one float4 load per tap.  Block totals: DPP row sums (4 VALU
// ops per accumulator) + one LDS hop, instead of a 6-step ds_bpermute butterfly per accumulator.
template <int CP, int KS>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* x, long ldx, const float* dy, long lddy, int B, int H,
                                                    int W, int Cin, int Cout, int pad, int dil, int vec4,
                                                    long pix_per_block, float* partial) {
  constexpr int T = KS * KS, NA = CP * T + 1;
  __shared__ float red[16][NA];
  const int n = blockIdx.y;
  const long total = (long)B * H * W;
  const long p0 = blockIdx.x * pix_per_block, p1 = min(total, p0 + pix_per_block);
  float acc[CP][T], bs = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[c][t] = 0.f;
  for (long pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    const int xx = pix % W;
    const long q = pix / W;
    const int yy = q % H;
    const long b = q / H;
    const float g = dy[pix * lddy + n];
    bs += g;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int sy = yy + ky * dil - pad;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int sx = xx + kx * dil - pad;
        if (sx < 0 || sx >= W) continue;
        const float* src = x + ((b * H + sy) * W + sx) * ldx;
        if (CP == 4 && vec4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][ky * KS + kx] += g * v[c];
        } else {
#pragma unroll
          for (int c = 0; c < CP; ++c)
            if (c < Cin) acc[c][ky * KS + kx] += g * src[c];
        }
      }
    }
  }
  auto row16 = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
  };
  const int rowid = threadIdx.x >> 4;
  const bool lead = (threadIdx.x & 15) == 0;
#pragma unroll
  for (int c = 0; c < CP; ++c)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float sum = row16(acc[c][t]);
      if (lead) red[rowid][c * T + t] = sum;
    }
  bs = row16(bs);
  if (lead) red[rowid][CP * T] = bs;
  __syncthreads();
  float* out = partial + ((long)blockIdx.x * Cout + n) * (Cin * T + 1);
  for (int i = threadIdx.x; i < Cin * T + 1; i += 256) {
    const int src = i == Cin * T ? CP * T : i;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += red[r][src];
    out[i] = sum;
  }
}

// The same with ALL output channels in one pass (round 5): wgrad_kernel runs one grid slice per output channel, i.e. reads x
// Cout times (the 3 x 3 radar_projection 4 -> 3 at 512 x 512 x 8: 3 x 33 MB through nine taps, 101 us = 0.6 TB/s on the tail of
// the backward pass); here a thread keeps NO x CP x T accumulators (108 for that layer) and x, dy are read once.
template <int CP, int KS, int NO>
__global__ __launch_bounds__(256) void wgrad_all_kernel(const float* x, long ldx, const float* dy, long lddy, int B, int H,
                                                        int W, int Cin, int pad, int dil, int vec4, long pix_per_block,
                                                        float* partial) {
  constexpr int T = KS * KS, NA = CP * T + 1;
  __shared__ float red[16][NO * NA];
  const long total = (long)B * H * W;
  const long p0 = blockIdx.x * pix_per_block, p1 = min(total, p0 + pix_per_block);
  float acc[NO][CP][T], bs[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    bs[n] = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[n][c][t] = 0.f;
  }
  for (long pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    const int xx = pix % W;
    const long q = pix / W;
    const int yy = q % H;
    float g[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      g[n] = dy[pix * lddy + n];
      bs[n] += g[n];
    }
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int sy = yy + ky * dil - pad;
      if (KS > 1 && (sy < 0 || sy >= H)) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int sx = xx + kx * dil - pad;
        if (KS > 1 && (sx < 0 || sx >= W)) continue;
        const float* src = x + (pix + (long)(sy - yy) * W + (sx - xx)) * ldx;
        float v[CP];
        if (CP == 4 && vec4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = t4[c];
        } else {
#pragma unroll
          for (int c = 0; c < CP; ++c) v[c] = c < Cin ? src[c] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NO; ++n)
#pragma unroll
          for (int c = 0; c < CP; ++c) acc[n][c][ky * KS + kx] += g[n] * v[c];
      }
    }
  }
  auto row16 = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
  };
  const int rowid = threadIdx.x >> 4;
  const bool lead = (threadIdx.x & 15) == 0;
#pragma unroll
  for (int n = 0; n < NO; ++n) {
#pragma unroll
    for (int c = 0; c < CP; ++c)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float sum = row16(acc[n][c][t]);
        if (lead) red[rowid][n * NA + c * T + t] = sum;
      }
    const float b = row16(bs[n]);
    if (lead) red[rowid][n * NA + CP * T] = b;
  }
  __syncthreads();
  const int per = Cin * T + 1;
  for (int i = threadIdx.x; i < NO * per; i += 256) {
    const int n = i / per, j = i - n * per;
    const int src = n * NA + (j == Cin * T ? CP * T : j);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += red[r][src];
    partial[((long)blockIdx.x * NO + n) * per + j] = sum;
  }
}

// dw (OIHW) [n][c][t] and db[n]; 16 lanes per output, blocks strided over the lanes, added in lane order via LDS
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, int nblk, int Cin, int Cout, int T,
                                                           const float* row_scale, float* dw, float* db, int accumulate) {
  __shared__ double red[16][16];
  const int per = Cin * T + 1;
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + o;
  const bool live = e < Cout * per;
  const int n = live ? e / per : 0, i = e - n * per;
  double s = 0;
  if (live)
    for (int b = sl; b < nblk; b += 16) s += partial[((long)b * Cout + n) * per + i];
  red[sl][o] = s;
  __syncthreads();
  if (sl != 0 || !live) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][o];
  float v = (float)s;
  if (row_scale) v *= row_scale[n];
  if (i == Cin * T) {
    if (db) db[n] = accumulate ? db[n] + v : v;
  } else {
    float* d = dw + (long)n * Cin * T + i;
    *d = accumulate ? *d + v : v;
  }
}

// Data gradient of a non-overlapping patch embedding (k == stride, pad 0):
// dx[b, y, x, c] = sum_n dy[b, y/k, x/k, n] * w[(y%k)*k + x%k][n][c]
__global__ __launch_bounds__(256) void patch_dgrad_kernel(const float* dy, long lddy, const float* w, float* dx, long lddx,
                                                          int B, int H, int W, int Cin, int Cout, int k, int accumulate) {
  extern __shared__ float wsm[];   // [T][Cout][Cin]
  const int T = k * k;
  for (int i = threadIdx.x; i < T * Cout * Cin; i += 256) wsm[i] = w[i];
  __syncthreads();
  const int OW = W / k, OH = H / k;
  const long total = (long)B * H * W;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int x = pix % W;
    const long q = pix / W;
    const int y = q % H;
    const long b = q / H;
    const int t = (y % k) * k + (x % k);
    const float* g = dy + ((b * OH + y / k) * OW + x / k) * lddy;
    const float* wt = wsm + (long)t * Cout * Cin;
    float acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
    for (int n = 0; n < Cout; ++n) {
      const float gv = g[n];
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < Cin) acc[c] += gv * wt[n * Cin + c];
    }
    float* d = dx + pix * lddx;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < Cin) d[c] = accumulate ? d[c] + acc[c] : acc[c];
  }
}

}  // namespace tiny

// Internal entry points used by vrnet_conv2d_f32 / vrnet_conv2d_wgrad_f32 (igemm.hip) for tiny-channel layers.
int vr_tiny_conv(int mode, const float* a, long lda, const float* w, const float* bias, float* y, long ldy, int B, int H,
                 int W, int Cin, int Cout, int k, int pad, int dil, int accumulate, hipStream_t st) {
  tiny::Args p{a, lda, w, bias, y, ldy, B, H, W, Cin, Cout, k, pad, dil, accumulate};
  long blocks = vr_cdiv((long)B * H * W, 256);
  if (blocks > 16384) blocks = 16384;
  // the shapes of the input fusion: fully unrolled kernels (a 4-channel row must then be 16-byte aligned when its stride is 4)
  const bool al = (lda != 4 || vr_aligned16(a)) && (ldy != 4 || vr_aligned16(y)) && (long)B * H * W < (1L << 31);
#define VR_TINY_FIXED(MODE_, CK_, CN_, KS_)                                                                        \
  do {                                                                                                             \
    hipLaunchKernelGGL((tiny::conv_fixed_kernel<MODE_, CK_, CN_, KS_>), dim3(blocks), dim3(256), 0, st, p);          \
    VR_LAUNCH_CHECK("tiny_conv");                                                                                  \
    return VR_OK;                                                                                                  \
  } while (0)
  if (al && mode == 0) {
    if (Cin == 3 && Cout == 3 && k == 1) VR_TINY_FIXED(0, 3, 3, 1);
    if (Cin == 4 && Cout == 4 && k == 1) VR_TINY_FIXED(0, 4, 4, 1);
    if (Cin == 7 && Cout == 4 && k == 1) VR_TINY_FIXED(0, 7, 4, 1);
    if (Cin == 4 && Cout == 3 && k == 3) VR_TINY_FIXED(0, 4, 3, 3);
  } else if (al) {           // data gradient: contracts over Cout, produces Cin
    if (Cin == 7 && Cout == 4 && k == 1) VR_TINY_FIXED(1, 4, 7, 1);
    if (Cin == 4 && Cout == 3 && k == 3) VR_TINY_FIXED(1, 3, 4, 3);
    if (Cin == 4 && Cout == 4 && k == 1) VR_TINY_FIXED(1, 4, 4, 1);
    if (Cin == 3 && Cout == 3 && k == 1) VR_TINY_FIXED(1, 3, 3, 1);
  }
#undef VR_TINY_FIXED
  if (mode == 0) hipLaunchKernelGGL((tiny::conv_kernel<0>), dim3(blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((tiny::conv_kernel<1>), dim3(blocks), dim3(256), 0, st, p);
  VR_LAUNCH_CHECK("tiny_conv");
  return VR_OK;
}

static void tiny_wgrad_plan(long npix, int* nblk, long* ppb) {
  // ~4 pixels per thread (round 5: at 32 per thread the 256 workgroups of the 512 x 512 x 8 launches left every CU with one
  // workgroup walking dependent loads: 100 us for 60 MB)
  long nb = vr_cdiv(npix, 1024);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  *ppb = vr_cdiv(npix, nb);
  *nblk = (int)vr_cdiv(npix, *ppb);
}

long vr_tiny_wgrad_workspace(long npix, int Cin, int Cout, int T) {
  int nblk;
  long ppb;
  tiny_wgrad_plan(npix, &nblk, &ppb);
  return (long)nblk * Cout * (Cin * T + 1) * 4 + 256;
}

int vr_tiny_wgrad(const float* x, long ldx, const float* dy, long lddy, float* dw, float* db, const float* row_scale,
                  int B, int H, int W, int Cin, int Cout, int k, int pad, int dil, int accumulate, void* workspace,
                  hipStream_t st) {
  int nblk;
  long ppb;
  tiny_wgrad_plan((long)B * H * W, &nblk, &ppb);
  float* partial = reinterpret_cast<float*>(workspace);
  const int vec4 = Cin == 4 && ldx % 4 == 0 && vr_aligned16(x);
  dim3 grid(nblk, Cout), block(256);
  // every output channel in ONE pass over x and dy for the layer shapes of the input fusion
#define VR_TINY_WGRAD_ALL(CP_, KS_, NO_)                                                                                 \
  do {                                                                                                                   \
    hipLaunchKernelGGL((tiny::wgrad_all_kernel<CP_, KS_, NO_>), dim3(nblk), block, 0, st, x, ldx, dy, lddy, B, H, W, Cin, pad, \
                       dil, vec4, ppb, partial);                                                                         \
    goto reduce;                                                                                                         \
  } while (0)
  if (k == 3 && Cin == 4 && Cout == 3) VR_TINY_WGRAD_ALL(4, 3, 3);
  if (k == 1 && Cin == 7 && Cout == 4) VR_TINY_WGRAD_ALL(8, 1, 4);
  if (k == 1 && Cin == 4 && Cout == 4) VR_TINY_WGRAD_ALL(4, 1, 4);
  if (k == 1 && Cin == 3 && Cout == 3) VR_TINY_WGRAD_ALL(4, 1, 3);
#undef VR_TINY_WGRAD_ALL
#define VR_TINY_WGRAD(CP_, KS_)                                                                                          \
  hipLaunchKernelGGL((tiny::wgrad_kernel<CP_, KS_>), grid, block, 0, st, x, ldx, dy, lddy, B, H, W, Cin, Cout, pad, dil, vec4, \
                     ppb, partial)
  if (k == 1) {
    if (Cin <= 4) VR_TINY_WGRAD(4, 1);
    else VR_TINY_WGRAD(8, 1);
  } else {
    if (Cin <= 4) VR_TINY_WGRAD(4, 3);
    else VR_TINY_WGRAD(8, 3);
  }
#undef VR_TINY_WGRAD
reduce:
  VR_LAUNCH_CHECK("tiny_wgrad");
  const int per = Cin * k * k + 1;
  hipLaunchKernelGGL(tiny::wgrad_reduce_kernel, dim3(vr_cdiv(Cout * per, 16)), dim3(256), 0, st, partial, nblk, Cin, Cout,
                     k * k, row_scale, dw, db, accumulate);
  VR_LAUNCH_CHECK("tiny_wgrad_reduce");
  return VR_OK;
}

int vr_patch_dgrad(const float* dy, long lddy, const float* w, float* dx, long lddx, int B, int H, int W, int Cin,
                   int Cout, int k, int accumulate, hipStream_t st) {
  long blocks = vr_cdiv((long)B * H * W, 256);
  if (blocks > 16384) blocks = 16384;
  const size_t lds = (size_t)k * k * Cout * Cin * sizeof(float);
  hipLaunchKernelGGL(tiny::patch_dgrad_kernel, dim3(blocks), dim3(256), lds, st, dy, lddy, w, dx, lddx, B, H, W, Cin, Cout,
                     k, accumulate);
  VR_LAUNCH_CHECK("patch_dgrad");
  return VR_OK;
}
