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

// partial[blk][n][Cin*T + 1]: sum over this block's pixels of dy[pix][n] * x[pix + off_t][c], last = sum dy (bias)
__global__ __launch_bounds__(256) void wgrad_kernel(const float* x, long ldx, const float* dy, long lddy, int B, int H,
                                                    int W, int Cin, int Cout, int k, int pad, int dil,
                                                    long pix_per_block, float* partial) {
  __shared__ float red[4][CMAX * 9 + 1];
  const int n = blockIdx.y, T = k * k;
  const long total = (long)B * H * W;
  const long p0 = blockIdx.x * pix_per_block, p1 = min(total, p0 + pix_per_block);
  float acc[CMAX][9], bs = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  for (long pix = p0 + threadIdx.x; pix < p1; pix += 256) {
    const int xx = pix % W;
    const long q = pix / W;
    const int yy = q % H;
    const long b = q / H;
    const float g = dy[pix * lddy + n];
    bs += g;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (ky >= k) break;
      const int sy = yy + ky * dil - pad;
      if (sy < 0 || sy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (kx >= k) break;
        const int sx = xx + kx * dil - pad;
        if (sx < 0 || sx >= W) continue;
        const float* src = x + ((b * H + sy) * W + sx) * ldx;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < Cin) acc[c][ky * 3 + kx] += g * src[c];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float s = wave_sum(acc[c][t]);
      if (lane == 0) red[wave][c * 9 + t] = s;
    }
  bs = wave_sum(bs);
  if (lane == 0) red[wave][CMAX * 9] = bs;
  __syncthreads();
  float* out = partial + ((long)blockIdx.x * Cout + n) * (Cin * T + 1);
  for (int i = threadIdx.x; i < Cin * T + 1; i += 256) {
    int src;
    if (i == Cin * T) src = CMAX * 9;
    else {
      const int c = i / T, t = i - c * T;
      src = c * 9 + (t / k) * 3 + (t % k);
    }
    out[i] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
  }
}

// dw (OIHW) [n][c][t] and db[n]
__global__ void wgrad_reduce_kernel(const float* partial, int nblk, int Cin, int Cout, int T, const float* row_scale,
                                    float* dw, float* db, int accumulate) {
  const int per = Cin * T + 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cout * per) return;
  const int n = e / per, i = e - n * per;
  double s = 0;
  for (int b = 0; b < nblk; ++b) s += partial[((long)b * Cout + n) * per + i];
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
  if (mode == 0) hipLaunchKernelGGL((tiny::conv_kernel<0>), dim3(blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((tiny::conv_kernel<1>), dim3(blocks), dim3(256), 0, st, p);
  VR_LAUNCH_CHECK("tiny_conv");
  return VR_OK;
}

static void tiny_wgrad_plan(long npix, int* nblk, long* ppb) {
  long nb = vr_cdiv(npix, 4096);
  if (nb > 1024) nb = 1024;
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
  hipLaunchKernelGGL(tiny::wgrad_kernel, dim3(nblk, Cout), dim3(256), 0, st, x, ldx, dy, lddy, B, H, W, Cin, Cout, k, pad,
                     dil, ppb, partial);
  VR_LAUNCH_CHECK("tiny_wgrad");
  const int per = Cin * k * k + 1;
  hipLaunchKernelGGL(tiny::wgrad_reduce_kernel, dim3(vr_cdiv(Cout * per, 128)), dim3(128), 0, st, partial, nblk, Cin, Cout,
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
