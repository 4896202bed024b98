// Spatial / attention streaming kernels of the fusion hot path (NHWC fp32):
//   depthwise 3x3 (head towers, backbone/conv_utils/normal_conv.py:23-33), bilinear x2/x4 upsampling with
//   align_corners=True (neck/coc_fpn_dual.py:19-22), the radar-guided image gain of ImageEnhanceByRadar with
//   its batch-global min/max normalisation (backbone/fusion/vr_coc.py:59-67,312-316) and ShuffleAttention
//   (backbone/attention_modules/shuffle_attention.py:48-72), each with its backward.
#include "common.h"

#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------ depthwise 3x3
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* x, long ldx, const float* w, float* y, long ldy,
                                                        int B, int H, int W, int C, int flip, int accumulate) {
  const int total = B * H * W * C;          // < 2^31, checked by the launcher: 32-bit index math
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c = e % C;
    const int pix = e / C;
    const int xx = pix % W;
    const int q = pix / W;
    const int yy = q % H;
    const int b = q / H;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int sy = yy + ky - 1;
      if (sy < 0 || sy >= H) continue;
      const float* row = x + ((long)(b * H + sy) * W) * ldx + c;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sx = xx + kx - 1;
        if (sx < 0 || sx >= W) continue;
        const int t = ky * 3 + kx;
        s += row[(long)sx * ldx] * w[c * 9 + (flip ? 8 - t : t)];
      }
    }
    float* d = y + (long)pix * ldy + c;
    *d = accumulate ? *d + s : s;
  }
}

// Vector variant (C % 4 == 0, C <= 1024, 16-byte rows): a thread owns 4 consecutive channels -- its 36 weights sit in
// registers for the whole kernel -- and walks pixels; every tap is one 16-byte load, a wave covers whole rows of
// channels.  The scalar kernel above issues 18 dword loads (9 of them 36-byte-strided weight reads) per output.
__global__ __launch_bounds__(256) void dwconv3x3_vec_kernel(const float* x, long ldx, const float* w, float* y, long ldy,
                                                            int B, int H, int W, int C, int flip, int accumulate,
                                                            int pix_per_block) {
  const int CQ = C >> 2, PP = 256 / CQ;          // channel quads, pixels in flight per pass (CQ divides 256 or PP = 1...)
  const int cq = threadIdx.x % CQ, pr = threadIdx.x / CQ;
  if (pr >= PP) return;
  f32x4 wt[9];
  {
    float raw[36];
    const float* ws = w + (long)cq * 36;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(ws + 4 * i);
      raw[4 * i] = t[0]; raw[4 * i + 1] = t[1]; raw[4 * i + 2] = t[2]; raw[4 * i + 3] = t[3];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ts = flip ? 8 - t : t;
      wt[t] = f32x4{raw[ts], raw[9 + ts], raw[18 + ts], raw[27 + ts]};
    }
  }
  const int total = B * H * W;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(total, p0 + pix_per_block);
  for (int pix = p0 + pr; pix < p1; pix += PP) {
    const int xx = pix % W;
    const int q = pix / W;
    const int yy = q % H;
    const int b = q / H;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int sy = yy + ky - 1;
      if (sy < 0 || sy >= H) continue;
      const float* row = x + ((long)(b * H + sy) * W) * ldx + 4 * cq;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int sx = xx + kx - 1;
        if (sx < 0 || sx >= W) continue;
        s += *reinterpret_cast<const f32x4*>(row + (long)sx * ldx) * wt[ky * 3 + kx];
      }
    }
    f32x4* d = reinterpret_cast<f32x4*>(y + (long)pix * ldy + 4 * cq);
    *d = accumulate ? *d + s : s;
  }
}

// Sliding-window variant: a thread owns 4 channels of a RUN of consecutive x positions of one image row and keeps the
// 3 x 3 window in registers: per output it loads ONE new column (3 x 16 bytes) instead of all nine taps, which is what
// bounds the kernel above (9 x 16 B through the vector L1 per 16 B written).
template <int RUN>
__global__ __launch_bounds__(256) void dwconv3x3_slide_kernel(const float* x, long ldx, const float* w, float* y, long ldy,
                                                              int B, int H, int W, int C, int flip, int accumulate) {
  const int CQ = C >> 2;
  const long item = (long)blockIdx.x * 256 + threadIdx.x;
  const int cq = (int)(item % CQ);
  const long run = item / CQ;
  const int runs_per_row = W / RUN;
  const long nruns = (long)B * H * runs_per_row;
  if (run >= nruns) return;
  const int xr = (int)(run % runs_per_row);
  const long rowi = run / runs_per_row;                 // b * H + yy
  const int yy = (int)(rowi % H);
  f32x4 wt[9];
  {
    float raw[36];
    const float* ws = w + (long)cq * 36;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(ws + 4 * i);
      raw[4 * i] = t[0]; raw[4 * i + 1] = t[1]; raw[4 * i + 2] = t[2]; raw[4 * i + 3] = t[3];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ts = flip ? 8 - t : t;
      wt[t] = f32x4{raw[ts], raw[9 + ts], raw[18 + ts], raw[27 + ts]};
    }
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float* rows[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int sy = yy + ky - 1;
    rows[ky] = (sy >= 0 && sy < H) ? x + ((rowi + ky - 1) * W) * ldx + 4 * cq : nullptr;
  }
  const int x0 = xr * RUN;
  auto column = [&](int sx, f32x4 (&col)[3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
      col[ky] = (rows[ky] != nullptr && sx >= 0 && sx < W) ? *reinterpret_cast<const f32x4*>(rows[ky] + (long)sx * ldx) : zero;
  };
  f32x4 c0[3], c1[3], c2[3];
  column(x0 - 1, c0);
  column(x0, c1);
#pragma unroll
  for (int i = 0; i < RUN; ++i) {
    column(x0 + i + 1, c2);
    f32x4 sacc = zero;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) sacc += c0[ky] * wt[ky * 3] + c1[ky] * wt[ky * 3 + 1] + c2[ky] * wt[ky * 3 + 2];
    f32x4* d = reinterpret_cast<f32x4*>(y + ((rowi * W) + x0 + i) * ldy + 4 * cq);
    *d = accumulate ? *d + sacc : sacc;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; }
  }
}

// partial[chunk][c][9]: dw[c][t] = sum_pix dy[pix, c] * x[pix + tap t, c].  A chunk is a run of image rows
// (b, y); all index math is 32-bit and row-uniform.
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* x, long ldx, const float* dy, long lddy,
                                                              int B, int H, int W, int C, long rows_per_chunk,
                                                              float* partial) {
  __shared__ float sm[256 * 9];
  const int TPR = C <= 32 ? 32 : 64;        // 4-8 x positions in flight per channel: more workgroups, more ILP
  const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR, RP = 256 / TPR;
  const int c = blockIdx.y * TPR + tx;
  const int nrows = B * H;
  const int r0 = (int)(blockIdx.x * rows_per_chunk), r1 = min(nrows, r0 + (int)rows_per_chunk);
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (c < C) {
    for (int row = r0; row < r1; ++row) {
      const int b = row / H, yy = row - b * H;
      for (int xx = ty; xx < W; xx += RP) {
        const float g = dy[((long)row * W + xx) * lddy + c];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int sy = yy + ky - 1;
          if (sy < 0 || sy >= H) continue;
          const float* src = x + ((long)(b * H + sy) * W) * ldx + c;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int sx = xx + kx - 1;
            if (sx < 0 || sx >= W) continue;
            acc[ky * 3 + kx] += g * src[(long)sx * ldx];
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) sm[threadIdx.x * 9 + t] = acc[t];
  __syncthreads();
  if (ty == 0 && c < C) {
    for (int r = 1; r < RP; ++r)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] += sm[(r * TPR + tx) * 9 + t];
#pragma unroll
    for (int t = 0; t < 9; ++t) partial[((long)blockIdx.x * C + c) * 9 + t] = acc[t];
  }
}
// Sliding-window form for the vector-aligned maps (C % 4 == 0, W % 16 == 0: every depthwise conv of the head): a thread owns
// four channels and a 16-pixel run of an image row and carries the 3 x 3 input window along it in registers -- per pixel
// three new 16-byte loads of x and one of dy instead of nine scalar loads of x and one of dy (the kernel above reads
// every input element nine times through the cache and streamed at 1.1 TB/s); the three rows a window spans are re-read by
// the thread's next row from L2.  The threads of a channel quad (row runs x rows of the workgroup) meet through LDS in a
// fixed order; one partial per workgroup and channel, reduced as before.
__device__ __forceinline__ f32x4 dw_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_slide_kernel(const float* x, long ldx, const float* dy, long lddy,
                                                                    int B, int H, int W, int C, int rows_per_chunk,
                                                                    float* partial) {
  constexpr int SEG = 16;
  extern __shared__ __attribute__((aligned(16))) float smx[];       // [groups - 1][quads][9] float4
  const int quads = min(C / 4, 64), nseg = W / SEG;
  const int cq = threadIdx.x % quads, rest = threadIdx.x / quads;      // rest: (row lane, segment)
  const int groups = 256 / quads;
  // nsegp of a row's 16-pixel runs are in flight at once (all of them where they fit the workgroup's groups; wider rows --
  // 128-pixel maps at 1 024 px -- are walked in steps of nsegp), rpar rows side by side
  const int nsegp = min(nseg, groups);
  const int seg = rest % nsegp, rl = rest / nsegp, rpar = groups / nsegp;
  const int c = (blockIdx.y * quads + cq) * 4;
  const int nrows = B * H;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(nrows, r0 + rows_per_chunk);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero;
  const bool live = c < C && rl < rpar;
  if (live) {
    for (int row = r0 + rl; row < r1; row += rpar)
    for (int sg = seg; sg < nseg; sg += nsegp) {
      const int x0 = sg * SEG;
      const int b = row / H, yy = row - b * H;
      const float* xr[3];
      bool ok[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int sy = yy + ky - 1;
        ok[ky] = sy >= 0 && sy < H;
        xr[ky] = x + ((long)(b * H + (ok[ky] ? sy : yy)) * W) * ldx + c;
      }
      const float* gr = dy + ((long)row * W) * lddy + c;
      f32x4 w0[3], w1[3], w2[3];      // window columns xx - 1, xx, xx + 1
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        w0[ky] = (ok[ky] && x0 > 0) ? dw_ld4(xr[ky] + (long)(x0 - 1) * ldx) : zero;
        w1[ky] = ok[ky] ? dw_ld4(xr[ky] + (long)x0 * ldx) : zero;
      }
#pragma unroll 4
      for (int i = 0; i < SEG; ++i) {
        const int xx = x0 + i;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) w2[ky] = (ok[ky] && xx + 1 < W) ? dw_ld4(xr[ky] + (long)(xx + 1) * ldx) : zero;
        const f32x4 g = dw_ld4(gr + (long)xx * lddy);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc[ky * 3] += g * w0[ky];
          acc[ky * 3 + 1] += g * w1[ky];
          acc[ky * 3 + 2] += g * w2[ky];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { w0[ky] = w1[ky]; w1[ky] = w2[ky]; }
      }
    }
  }
  if (rest > 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(smx + (((rest - 1) * quads + cq) * 9 + t) * 4) = acc[t];
  }
  __syncthreads();
  if (rest == 0 && c < C) {
    for (int r = 1; r < groups; ++r)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] += *reinterpret_cast<const f32x4*>(smx + (((r - 1) * quads + cq) * 9 + t) * 4);
    float* d = partial + ((long)blockIdx.x * C + c) * 9;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) d[j * 9 + t] = acc[t][j];
  }
}
// 16 lanes per output, chunks strided over the lanes, partials added in lane order through LDS (deterministic)
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_reduce_kernel(const float* partial, int nchunks, int C, float* dw,
                                                                     int accumulate) {
  __shared__ double red[16][16];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + o;
  const bool live = e < C * 9;
  double s = 0;
  if (live)
    for (int k = sl; k < nchunks; k += 16) s += partial[(long)k * C * 9 + e];
  red[sl][o] = s;
  __syncthreads();
  if (sl == 0 && live) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][o];
    dw[e] = (accumulate ? dw[e] : 0.f) + (float)s;
  }
}

// ------------------------------------------------------------------------------------------ patch embedding
// Non-overlapping k x k / stride-k patch embedding (PointRecuder k4 s4, vr_coc.py:83-102, fed by
// cat([x, fea_pos]) :583-586) as gather + plain GEMM: P[b, oy, ox, (ky*k + kx)*CT + c] = cat(x, pos)[b, oy*k+ky,
// ox*k+kx, c], CT = C + CP.  The concat never materialises; the 1x1 conv over P (weights in OHWI order) then runs
// on the vector path of the implicit GEMM with K = k*k*CT, instead of gathering 5- / 6-channel rows per tap.
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* x, long ldx, const float* pos, float* out, int B,
                                                           int H, int W, int C, int CP, int k) {
  const int CT = C + CP, KT = k * k * CT, OH = H / k, OW = W / k;
  const long total = (long)B * OH * OW * KT;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int ch = e % KT;
    const long op = e / KT;
    const int t = ch / CT, c = ch - t * CT;
    const int ky = t / k, kx = t - ky * k;
    const int ox = op % OW;
    const long q = op / OW;
    const int oy = q % OH;
    const long b = q / OH;
    const int yy = oy * k + ky, xx = ox * k + kx;
    out[e] = c < C ? x[((b * H + yy) * W + xx) * ldx + c] : pos[((long)yy * W + xx) * CP + (c - C)];
  }
}
// adjoint w.r.t. x: dx[b, y, x, c] (+)= dP[b, y/k, x/k, ((y%k)*k + x%k)*CT + c], c < C
__global__ __launch_bounds__(256) void patch_scatter_kernel(const float* dp, float* dx, long lddx, int B, int H, int W, int C,
                                                            int CP, int k, int accumulate) {
  const int CT = C + CP, KT = k * k * CT, OH = H / k, OW = W / k;
  const long total = (long)B * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = e % C;
    const long pix = e / C;
    const int xx = pix % W;
    const long q = pix / W;
    const int yy = q % H;
    const long b = q / H;
    const int t = (yy % k) * k + (xx % k);
    const float v = dp[((b * OH + yy / k) * OW + xx / k) * KT + t * CT + c];
    float* d = dx + pix * lddx + c;
    *d = accumulate ? *d + v : v;
  }
}
// weights OIHW [n][c][t] -> OHWI [n][t][c] (dir 0), and the gradient back, OHWI -> OIHW (+)= (dir 1)
__global__ void ohwi_kernel(const float* src, float* dst, int Cout, int Cin, int T, int dir, int accumulate) {
  const long total = (long)Cout * Cin * T;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int t = e % T;
  const long q = e / T;
  const int c = q % Cin;
  const long n = q / Cin;
  const long oihw = e, ohwi = (n * T + t) * Cin + c;
  if (dir == 0) dst[ohwi] = src[oihw];
  else dst[oihw] = (accumulate ? dst[oihw] : 0.f) + src[ohwi];
}

// ------------------------------------------------------------------------------------------ bilinear upsample
__device__ __forceinline__ void src_index(float scale, int o, int in_size, int* i0, int* i1, float* l1) {
  const float r = scale * (float)o;
  int a = (int)r;
  if (a > in_size - 1) a = in_size - 1;
  *i0 = a;
  *i1 = a + ((a < in_size - 1) ? 1 : 0);
  *l1 = r - (float)a;
}

// bnA != nullptr (round 5, K11): the taps are ReLU(BatchNorm(z)) of the pre-normalisation map x = z, evaluated on the fly as
// fmaxf(fma(A[c], z - S[c], D[c]), 0) -- the expression of the apply kernel (stream_ops.hip bn_pre), so the result has the bits of
// affine + upsample while the low-resolution activation is never stored.
__device__ __forceinline__ float up_tap(float v, const float* bnA, const float* bnD, const float* bnS, int c) {
  return bnA ? fmaxf(__builtin_fmaf(bnA[c], v - bnS[c], bnD[c]), 0.f) : v;
}

__global__ __launch_bounds__(256) void upsample_kernel(const float* x, long ldx, float* y, long ldy, int B, int H,
                                                       int W, int C, int OH, int OW, int out_nchw, const float* bnA,
                                                       const float* bnD, const float* bnS) {
  const float ry = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const long total = (long)B * OH * OW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int c, ox, oy;
    long b;
    if (out_nchw) {   // e runs over [b][c][oy][ox] so that stores coalesce
      ox = e % OW; long q = e / OW; oy = q % OH; q /= OH; c = q % C; b = q / C;
    } else {
      c = e % C; long q = e / C; ox = q % OW; q /= OW; oy = q % OH; b = q / OH;
    }
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(ry, oy, H, &y0, &y1, &ly);
    src_index(rx, ox, W, &x0, &x1, &lx);
    const float* base = x + (b * H * W) * ldx + c;
    const float v00 = up_tap(base[((long)y0 * W + x0) * ldx], bnA, bnD, bnS, c), v01 = up_tap(base[((long)y0 * W + x1) * ldx], bnA, bnD, bnS, c);
    const float v10 = up_tap(base[((long)y1 * W + x0) * ldx], bnA, bnD, bnS, c), v11 = up_tap(base[((long)y1 * W + x1) * ldx], bnA, bnD, bnS, c);
    const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    if (out_nchw) y[e] = v;
    else y[((b * OH + oy) * OW + ox) * ldy + c] = v;
  }
}

// NHWC output on 16-byte accesses: a thread interpolates four consecutive channels of one output pixel (index arithmetic
// once per quad; the same expression per component as the scalar kernel).
__global__ __launch_bounds__(256) void upsample_vec_kernel(const float* x, long ldx, float* y, long ldy, int B, int H, int W,
                                                           int C, int OH, int OW, const float* bnA, const float* bnD,
                                                           const float* bnS) {
  const float ry = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int CQ = C >> 2;
  const long total = (long)B * OH * OW * CQ;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int cq = e % CQ;
    long q = e / CQ;
    const int ox = q % OW; q /= OW;
    const int oy = q % OH;
    const long b = q / OH;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(ry, oy, H, &y0, &y1, &ly);
    src_index(rx, ox, W, &x0, &x1, &lx);
    const float* base = x + (b * H * W) * ldx + 4 * cq;
    f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x0) * ldx);
    f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long)y0 * W + x1) * ldx);
    f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x0) * ldx);
    f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long)y1 * W + x1) * ldx);
    if (bnA) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * cq + j;
        v00[j] = up_tap(v00[j], bnA, bnD, bnS, c);
        v01[j] = up_tap(v01[j], bnA, bnD, bnS, c);
        v10[j] = up_tap(v10[j], bnA, bnD, bnS, c);
        v11[j] = up_tap(v11[j], bnA, bnD, bnS, c);
      }
    }
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (1.f - ly) * ((1.f - lx) * v00[j] + lx * v01[j]) + ly * ((1.f - lx) * v10[j] + lx * v11[j]);
    *reinterpret_cast<f32x4*>(y + ((b * OH + oy) * OW + ox) * ldy + 4 * cq) = v;
  }
}

// The adjoint for NHWC dy on 16-byte accesses (same gather, four channels per thread).
__global__ __launch_bounds__(256) void upsample_bwd_vec_kernel(const float* dy, long lddy, float* dx, long lddx, int B, int H,
                                                               int W, int C, int OH, int OW, int accumulate) {
  const float ry = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int CQ = C >> 2;
  const long total = (long)B * H * W * CQ;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int cq = e % CQ;
    long q = e / CQ;
    const int ix = q % W; q /= W;
    const int iy = q % H;
    const long b = q / H;
    int oy_lo = 0, oy_hi = OH - 1, ox_lo = 0, ox_hi = OW - 1;
    if (ry > 0.f) {
      oy_lo = max(0, (int)floorf((float)(iy - 1) / ry) - 1);
      oy_hi = min(OH - 1, (int)ceilf((float)(iy + 1) / ry) + 1);
    }
    if (rx > 0.f) {
      ox_lo = max(0, (int)floorf((float)(ix - 1) / rx) - 1);
      ox_hi = min(OW - 1, (int)ceilf((float)(ix + 1) / rx) + 1);
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      src_index(ry, oy, H, &y0, &y1, &ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        src_index(rx, ox, W, &x0, &x1, &lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((b * OH + oy) * (long)OW + ox) * lddy + 4 * cq);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += w * g[j];
      }
    }
    f32x4* d = reinterpret_cast<f32x4*>(dx + ((b * H + iy) * (long)W + ix) * lddx + 4 * cq);
    if (accumulate) s += *d;
    *d = s;
  }
}

// NCHW output (the seg logits, C = num_seg_classes): a thread writes four consecutive ox of one (b, c, oy) row as one 16-byte
// store; the row pair and its weight are found once per thread.  Same expression per output as upsample_kernel.
__global__ __launch_bounds__(256) void upsample_nchw4_kernel(const float* x, long ldx, float* y, int B, int H, int W, int C,
                                                             int OH, int OW, const float* bnA, const float* bnD,
                                                             const float* bnS) {
  const float ry = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int OQ = OW >> 2;
  const long total = (long)B * C * OH * OQ;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int oq = e % OQ;
    long q = e / OQ;
    const int oy = q % OH; q /= OH;
    const int c = q % C;
    const long b = q / C;
    int y0, y1;
    float ly;
    src_index(ry, oy, H, &y0, &y1, &ly);
    const float* r0 = x + ((b * H + y0) * (long)W) * ldx + c;
    const float* r1 = x + ((b * H + y1) * (long)W) * ldx + c;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      src_index(rx, 4 * oq + j, W, &x0, &x1, &lx);
      const float v00 = up_tap(r0[(long)x0 * ldx], bnA, bnD, bnS, c), v01 = up_tap(r0[(long)x1 * ldx], bnA, bnD, bnS, c);
      const float v10 = up_tap(r1[(long)x0 * ldx], bnA, bnD, bnS, c), v11 = up_tap(r1[(long)x1 * ldx], bnA, bnD, bnS, c);
      v[j] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
    *reinterpret_cast<f32x4*>(y + ((b * C + c) * OH + oy) * (long)OW + 4 * oq) = v;
  }
}

// gather form of the adjoint: deterministic, no atomics
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* dy, long lddy, int dy_nchw, float* dx, long lddx,
                                                           int B, int H, int W, int C, int OH, int OW, int accumulate) {
  const float ry = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const long total = (long)B * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int c, ix, iy;
    long b;
    if (dy_nchw) {      // neighbouring threads = neighbouring ix of one channel plane: their dy reads share lines (round 4: with
      ix = e % W;       // c fastest they came from planes megabytes apart -- 0.56 TB/s on the seg logits' gradient)
      long q = e / W;
      iy = q % H; q /= H;
      c = q % C;
      b = q / C;
    } else {
      c = e % C;
      long q = e / C;
      ix = q % W; q /= W;
      iy = q % H;
      b = q / H;
    }
    int oy_lo = 0, oy_hi = OH - 1, ox_lo = 0, ox_hi = OW - 1;
    if (ry > 0.f) {
      oy_lo = max(0, (int)floorf((float)(iy - 1) / ry) - 1);
      oy_hi = min(OH - 1, (int)ceilf((float)(iy + 1) / ry) + 1);
    }
    if (rx > 0.f) {
      ox_lo = max(0, (int)floorf((float)(ix - 1) / rx) - 1);
      ox_hi = min(OW - 1, (int)ceilf((float)(ix + 1) / rx) + 1);
    }
    float s = 0.f;
    // the column weights do not depend on oy: found once (round 4; they were re-derived for every (oy, ox) pair -- 121 times
    // per element at scale 4); the terms and their order are unchanged
    constexpr int MAXR = 16;
    const int nx = ox_hi - ox_lo + 1;
    float wxs[MAXR];
    if (nx <= MAXR) {
#pragma unroll
      for (int t = 0; t < MAXR; ++t) {
        int x0, x1;
        float lx;
        src_index(rx, min(ox_lo + t, OW - 1), W, &x0, &x1, &lx);
        wxs[t] = t < nx ? (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f) : 0.f;
      }
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      src_index(ry, oy, H, &y0, &y1, &ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      if (nx <= MAXR) {
        const float* grow = dy_nchw ? dy + ((b * C + c) * OH + oy) * (long)OW + ox_lo : dy + ((b * OH + oy) * (long)OW + ox_lo) * lddy + c;
        const long gstep = dy_nchw ? 1 : lddy;
#pragma unroll
        for (int t = 0; t < MAXR; ++t) {
          const float wx = wxs[t];
          if (wx == 0.f) continue;
          s += wy * wx * grow[t * gstep];
        }
        continue;
      }
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        src_index(rx, ox, W, &x0, &x1, &lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        const float g = dy_nchw ? dy[((b * C + c) * OH + oy) * (long)OW + ox]
                                : dy[((b * OH + oy) * (long)OW + ox) * lddy + c];
        s += wy * wx * g;
      }
    }
    float* d = dx + ((b * H + iy) * (long)W + ix) * lddx + c;
    *d = accumulate ? *d + s : s;
  }
}

// ------------------------------------------------------------------------------------------ min / max, image gain
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* p, long n, float* partial) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float v = p[e];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
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
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    partial[2 * blockIdx.x + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* partial, int nblocks, float* mm) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
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
  if (threadIdx.x == 0) {
    mm[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mm[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

// out = (1 + (p - mn) / (mx - mn)) * x     (contiguous tensors of n elements)
__global__ void enhance_mul_kernel(const float* p, const float* x, const float* mm, float* out, long n) {
  const float mn = mm[0], dst = mm[1] - mm[0];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = (1.f + (p[e] - mn) / dst) * x[e];
}

// The same with the final step of the min / max reduction INSIDE (round 5): every workgroup folds the <= 512 partial pairs
// minmax_partial_kernel left (fminf / fmaxf: the same bits whatever the order) and workgroup 0 also leaves (min, max) in mm for
// the backward pass -- one launch less on the chain of every ImageEnhanceByRadar block.
__global__ __launch_bounds__(256) void enhance_fwd_kernel(const float* p, const float* x, const float* partial, int nblocks,
                                                          float* mm, float* out, long n, int vec) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
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
  if (blockIdx.x == 0 && threadIdx.x == 0) { mm[0] = mn; mm[1] = mx; }
  const float dst = mx - mn;
  if (vec) {
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(p + 4 * e), xv = *reinterpret_cast<const f32x4*>(x + 4 * e);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (1.f + (pv[j] - mn) / dst) * xv[j];
      *reinterpret_cast<f32x4*>(out + 4 * e) = o;
    }
  } else {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
      out[e] = (1.f + (p[e] - mn) / dst) * x[e];
  }
}

// sums: [0] sum dn, [1] sum dn*(p-mn), [2] #(p == mn), [3] #(p == mx)   with dn = dt * x
__global__ __launch_bounds__(256) void enhance_bwd_partial_kernel(const float* dt, const float* x, const float* p,
                                                                  const float* mm, long n, double* partial) {
  __shared__ double red[4][4];
  const float mn = mm[0], mx = mm[1];
  double s[4] = {0, 0, 0, 0};
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float pv = p[e];
    const double dn = (double)dt[e] * (double)x[e];
    s[0] += dn;
    s[1] += dn * (double)(pv - mn);
    s[2] += (pv == mn) ? 1.0 : 0.0;
    s[3] += (pv == mx) ? 1.0 : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = wave_sum(s[i]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[threadIdx.x >> 6][i] = s[i];
  __syncthreads();
  if (threadIdx.x < 4) partial[4 * (long)blockIdx.x + threadIdx.x] =
      red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// dx (+)= dt * (1 + n);  dp = dn/dst + [p==mn] g_mn/cnt_mn + [p==mx] g_mx/cnt_mx
__global__ __launch_bounds__(256) void enhance_bwd_apply_kernel(const float* dt, const float* x, const float* p, const float* mm,
                                                                const double* partial, int nblocks, float* dx, float* dp, long n,
                                                                int accumulate_dx) {
  // (round 5) the four sums are finished HERE: every workgroup adds the <= 256 partial quadruples in the same fixed order
  __shared__ double red[4][4];
  __shared__ double sums[4];
  {
    double s[4] = {0, 0, 0, 0};
    for (int e = threadIdx.x; e < nblocks; e += 256)
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += partial[4 * (long)e + i];
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
  const double dst = (double)mx - (double)mn;
  const float gmn = (float)((-sums[0] / dst + sums[1] / (dst * dst)) / sums[2]);
  const float gmx = (float)((-sums[1] / (dst * dst)) / sums[3]);
  const float fd = (float)dst;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const float pv = p[e], g = dt[e];
    const float gain = 1.f + (pv - mn) / fd;
    const float vx = g * gain;
    dx[e] = accumulate_dx ? dx[e] + vx : vx;
    float v = g * x[e] / fd;
    if (pv == mn) v += gmn;
    if (pv == mx) v += gmx;
    dp[e] = v;
  }
}

// ------------------------------------------------------------------------------------------ ShuffleAttention
struct SaParams {
  const float *cw, *cb, *sw, *sb, *gnw, *gnb;
};
__device__ __forceinline__ int sa_dst(int q, int C) { return q < C / 2 ? 2 * q : 2 * (q - C / 2) + 1; }

__global__ void sa_coef_fwd_kernel(const double* mom, SaParams sp, int B, long HW, int C, int G, float* P, float* Q,
                                   float* Mn) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C) return;
  const int q = e % C, cp = C / (2 * G);
  const int rem = q % (2 * cp), half = rem / cp, i = rem % cp;
  const double mean = mom[2 * e] / (double)HW;
  if (half == 0) {                       // gate = sigmoid(P * (x - Mn) + Q)
    P[e] = 0.f;
    Q[e] = (float)((double)sp.cw[i] * mean + (double)sp.cb[i]);
    Mn[e] = 0.f;
  } else {
    double var = mom[2 * e + 1] / (double)HW - mean * mean;
    if (var < 0) var = 0;
    const double r = 1.0 / sqrt(var + 1e-5);
    P[e] = (float)((double)sp.sw[i] * sp.gnw[i] * r);
    Q[e] = (float)((double)sp.sw[i] * (double)sp.gnb[i] + (double)sp.sb[i]);
    Mn[e] = (float)mean;
  }
}

__global__ __launch_bounds__(256) void sa_apply_kernel(const float* x, long ldx, const float* P, const float* Q,
                                                       const float* Mn, float* y, long ldy, long HW, int C) {
  const long b = blockIdx.y;
  const long total = HW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / C;
    const int q = e - r * C;
    const float xv = x[(b * HW + r) * ldx + q];
    const float z = P[b * C + q] * (xv - Mn[b * C + q]) + Q[b * C + q];
    y[(b * HW + r) * ldy + sa_dst(q, C)] = xv * vr_sigmoid(z);
  }
}

// per (b,q): T1 = sum dz, T2 = sum dz*x with dz = dy[dst(q)] * x * sig'(z); one workgroup per (row chunk, b)
__global__ __launch_bounds__(256) void sa_bwd_reduce_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                            const float* P, const float* Q, const float* Mn, long HW,
                                                            int C, int TPR, long rows_per_chunk, int nchunks,
                                                            double* partial) {
  extern __shared__ double smd[];   // [256][2]
  const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR, RP = 256 / TPR;
  const int chunk = blockIdx.x;
  const long b = blockIdx.y;
  const int q = blockIdx.z * TPR + tx;
  const long r0 = chunk * rows_per_chunk, r1 = min(HW, r0 + rows_per_chunk);
  double t1 = 0, t2 = 0;
  if (q < C) {
    const float pq = P[b * C + q], qq = Q[b * C + q], mq = Mn[b * C + q];
    const int dq = sa_dst(q, C);
    for (long r = r0 + ty; r < r1; r += RP) {
      const float xv = x[(b * HW + r) * ldx + q];
      const float sg = vr_sigmoid(pq * (xv - mq) + qq);
      const double dz = (double)(dy[(b * HW + r) * lddy + dq] * xv * sg * (1.f - sg));
      t1 += dz;
      t2 += dz * (double)xv;
    }
  }
  smd[threadIdx.x * 2] = t1;
  smd[threadIdx.x * 2 + 1] = t2;
  __syncthreads();
  if (ty == 0 && q < C) {
    for (int r = 1; r < RP; ++r) {
      t1 += smd[(r * TPR + tx) * 2];
      t2 += smd[(r * TPR + tx) * 2 + 1];
    }
    partial[((b * nchunks + chunk) * C + q) * 2] = t1;
    partial[((b * nchunks + chunk) * C + q) * 2 + 1] = t2;
  }
}
__global__ __launch_bounds__(256) void sa_reduce_chunks_kernel(const double* partial, double* out, int B, int nchunks, int C) {
  __shared__ double red[16][16];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + o;
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

// blocks [0, nb1): E,F per (b,q);  block nb1: parameter gradients (one thread per (half, i))
__global__ __launch_bounds__(256) void sa_coef_bwd_kernel(const double* T, const double* mom, SaParams sp, int B, long HW,
                                                          int C, int G, int nb1, float* E, float* F, float* dcw,
                                                          float* dcb, float* dsw, float* dsb, float* dgnw, float* dgnb,
                                                          int accumulate) {
  const int cp = C / (2 * G);
  if ((int)blockIdx.x < nb1) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * C) return;
    const int q = e % C;
    const int rem = q % (2 * cp), half = rem / cp, i = rem % cp;
    const double t1 = T[2 * e], t2 = T[2 * e + 1];
    const double mean = mom[2 * e] / (double)HW;
    if (half == 0) {
      E[e] = 0.f;
      F[e] = (float)((double)sp.cw[i] * t1 / (double)HW);
    } else {
      double var = mom[2 * e + 1] / (double)HW - mean * mean;
      if (var < 0) var = 0;
      const double r = 1.0 / sqrt(var + 1e-5);
      const double k = (double)sp.sw[i] * sp.gnw[i];
      const double m1 = k * t1 / (double)HW, m2 = k * r * (t2 - mean * t1) / (double)HW;
      E[e] = (float)(-r * r * m2);           // dx = dy * (...) + E * (x - Mn) + F
      F[e] = (float)(-r * m1);
    }
  } else {
    // one workgroup per parameter index i: the B * G (sample, group) terms are spread over the threads and added in a
    // fixed tree (was: one thread per i walking all B * G terms, a 50 us serial chain of dependent loads)
    __shared__ double red[6][4];
    const int i = (int)blockIdx.x - nb1;
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int pidx = threadIdx.x; pidx < B * G; pidx += 256) {
      const int b = pidx / G, g = pidx - b * G;
      {
        const long e = (long)b * C + g * 2 * cp + i;
        const double mean = mom[2 * e] / (double)HW;
        v[0] += mean * T[2 * e];
        v[1] += T[2 * e];
      }
      {
        const long e = (long)b * C + g * 2 * cp + cp + i;
        const double t1 = T[2 * e], t2 = T[2 * e + 1];
        const double mean = mom[2 * e] / (double)HW;
        double var = mom[2 * e + 1] / (double)HW - mean * mean;
        if (var < 0) var = 0;
        const double r = 1.0 / sqrt(var + 1e-5);
        const double xh = r * (t2 - mean * t1);   // sum dz * xhat
        v[2] += (double)sp.gnw[i] * xh + (double)sp.gnb[i] * t1;
        v[3] += t1;
        v[4] += (double)sp.sw[i] * xh;
        v[5] += (double)sp.sw[i] * t1;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double w = wave_sum(v[k]);
      if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float* outs[6] = {dcw, dcb, dsw, dsb, dgnw, dgnb};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double tot = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        outs[k][i] = (accumulate ? outs[k][i] : 0.f) + (float)tot;
      }
    }
  }
}

__global__ __launch_bounds__(256) void sa_bwd_apply_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                           const float* P, const float* Q, const float* Mn, const float* E,
                                                           const float* F, float* dx, long lddx, long HW, int C,
                                                           int accumulate) {
  const long b = blockIdx.y;
  const long total = HW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / C;
    const int q = e - r * C;
    const float xv = x[(b * HW + r) * ldx + q];
    const float pq = P[b * C + q], mq = Mn[b * C + q];
    const float sg = vr_sigmoid(pq * (xv - mq) + Q[b * C + q]);
    const float g = dy[(b * HW + r) * lddy + sa_dst(q, C)];
    const float v = g * (sg + xv * sg * (1.f - sg) * pq) + E[b * C + q] * (xv - mq) + F[b * C + q];
    float* d = dx + (b * HW + r) * lddx + q;
    *d = accumulate ? *d + v : v;
  }
}

inline long grid_for(long n, long per_block = 1024, long cap = 8192) {
  long g = vr_cdiv(n, per_block);
  return g < 1 ? 1 : (g > cap ? cap : g);
}

void sa_plan(long HW, int C, int* TPR, int* ncb, int* nchunks, long* rows) {
  int t = 1;
  while (t < C && t < 256) t <<= 1;
  *TPR = t;
  *ncb = (int)vr_cdiv(C, t);
  long nc = vr_cdiv(HW * C, 8192);
  if (nc < 1) nc = 1;
  if (nc > 1024) nc = 1024;
  if (nc > HW) nc = HW;
  *rows = vr_cdiv(HW, nc);
  *nchunks = (int)vr_cdiv(HW, *rows);
}

}  // namespace

extern "C" int vrnet_dwconv3x3_f32(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W,
                                   int C, int flip, int accumulate, void* stream) {
  if (vr_ablated("dwconv")) return VR_OK;
  VR_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && C > 0, "dwconv3x3: bad arguments");
  VR_CHECK_ARG((long)B * H * W * C < (1L << 31), "dwconv3x3: tensor too large");
  const bool vec = C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && ldx % 4 == 0 && ldy % 4 == 0 && vr_aligned16(x) &&
                   vr_aligned16(y) && vr_aligned16(w);
  static const int slide = vr_tune("VRNET_DW_SLIDE", 1);     // tuning aid
  if (vec && slide && W % 8 == 0) {
    const long items = (long)B * H * (W / 8) * (C / 4);
    hipLaunchKernelGGL((dwconv3x3_slide_kernel<8>), dim3(vr_cdiv(items, 256)), dim3(256), 0, vr_stream(stream), x, ldx, w, y,
                       ldy, B, H, W, C, flip, accumulate);
  } else if (vec) {
    const long npix = (long)B * H * W;
    const int pp = 256 / (C / 4);
    long ppb = pp * 8L;                              // >= 8 pixels per thread, at least ~2048 workgroups when possible
    while (vr_cdiv(npix, ppb) > 4096) ppb *= 2;
    hipLaunchKernelGGL(dwconv3x3_vec_kernel, dim3(vr_cdiv(npix, ppb)), dim3(256), 0, vr_stream(stream), x, ldx, w, y, ldy, B,
                       H, W, C, flip, accumulate, (int)ppb);
  } else {
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, vr_stream(stream), x, ldx, w, y,
                       ldy, B, H, W, C, flip, accumulate);
  }
  VR_LAUNCH_CHECK("dwconv3x3");
  return VR_OK;
}

static bool dw_wgrad_slide_ok(int W, int C, long ldx, long lddy, const void* x, const void* dy) {
  if (!(W % 16 == 0 && W <= 512 && C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && vr_aligned16(x) && vr_aligned16(dy))) return false;
  if (C / 4 > 64) return (C / 4) % 64 == 0 && (W / 16 <= 4 ? 4 % (W / 16) == 0 : (W / 16) % 4 == 0);
  if (256 % (C / 4) != 0) return false;
  const int groups = 256 / (C / 4), nseg = W / 16;      // runs of a row side by side, or the row in whole steps of `groups` runs
  return nseg <= groups ? groups % nseg == 0 : nseg % groups == 0;
}
static void dw_wgrad_plan(long nrows, int W, int C, int* nchunks, long* rpc) {   // chunks of whole image rows
  long nc = vr_cdiv(nrows * W * C, 16384);
  if (nc < 1) nc = 1;
  if (nc > 512) nc = 512;
  if (nc > nrows) nc = nrows;
  *rpc = vr_cdiv(nrows, nc);
  *nchunks = (int)vr_cdiv(nrows, *rpc);
}
extern "C" long vrnet_dwconv3x3_wgrad_workspace(int B, int H, int W, int C) {
  int nchunks;
  long ppc;
  dw_wgrad_plan((long)B * H, W, C, &nchunks, &ppc);
  return (long)nchunks * C * 9 * 4 + 256;
}
extern "C" int vrnet_dwconv3x3_wgrad_f32(const float* x, long ldx, const float* dy, long lddy, float* dw, int B, int H,
                                         int W, int C, int accumulate, void* workspace, long workspace_bytes,
                                         void* stream) {
  if (vr_ablated("dwconv")) return VR_OK;
  VR_CHECK_ARG(x && dy && dw && workspace, "dwconv3x3_wgrad: null tensor");
  int nchunks;
  long ppc;
  dw_wgrad_plan((long)B * H, W, C, &nchunks, &ppc);
  if (workspace_bytes < vrnet_dwconv3x3_wgrad_workspace(B, H, W, C)) {
    vr_set_error("dwconv3x3_wgrad: workspace too small");
    return VR_ERR_WORKSPACE;
  }
  const int TPR = C <= 32 ? 32 : 64;
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t st = vr_stream(stream);
  if (dw_wgrad_slide_ok(W, C, ldx, lddy, x, dy) && vr_tune("VRNET_DW_WGRAD_SLIDE", 1)) {
    // (the chunk plan above is in whole image rows with <= 512 chunks: <= 2.4 MB of partials at the head's largest map)
    const int quads = C / 4 < 64 ? C / 4 : 64, groups = 256 / quads;
    hipLaunchKernelGGL(dwconv3x3_wgrad_slide_kernel, dim3(nchunks, (C / 4) / quads), dim3(256), (size_t)(groups - 1) * quads * 9 * 16, st,
                       x, ldx, dy, lddy, B, H, W, C, (int)ppc, partial);
  } else {
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(nchunks, vr_cdiv(C, TPR)), dim3(256), 0, st, x, ldx, dy, lddy, B, H, W,
                       C, ppc, partial);
  }
  VR_LAUNCH_CHECK("dwconv3x3_wgrad");
  hipLaunchKernelGGL(dwconv3x3_wgrad_reduce_kernel, dim3(vr_cdiv(C * 9, 16)), dim3(256), 0, st, partial, nchunks, C, dw,
                     accumulate);
  VR_LAUNCH_CHECK("dwconv3x3_wgrad_reduce");
  return VR_OK;
}

static int upsample_launch(const float* x, long ldx, float* y, long ldy, int B, int H, int W, int C, int scale, int out_nchw,
                           const float* bnA, const float* bnD, const float* bnS, void* stream) {
  if (!out_nchw && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && vr_aligned16(x) && vr_aligned16(y))
    hipLaunchKernelGGL(upsample_vec_kernel, dim3(grid_for((long)B * H * W * scale * scale * (C / 4))), dim3(256), 0,
                       vr_stream(stream), x, ldx, y, ldy, B, H, W, C, H * scale, W * scale, bnA, bnD, bnS);
  else if (out_nchw && (W * scale) % 4 == 0 && vr_aligned16(y))
    hipLaunchKernelGGL(upsample_nchw4_kernel, dim3(grid_for((long)B * H * W * scale * scale * C / 4)), dim3(256), 0,
                       vr_stream(stream), x, ldx, y, B, H, W, C, H * scale, W * scale, bnA, bnD, bnS);
  else
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for((long)B * H * W * scale * scale * C)), dim3(256), 0,
                       vr_stream(stream), x, ldx, y, ldy, B, H, W, C, H * scale, W * scale, out_nchw, bnA, bnD, bnS);
  VR_LAUNCH_CHECK("upsample");
  return VR_OK;
}

extern "C" int vrnet_upsample_bilinear_f32(const float* x, long ldx, float* y, long ldy, int B, int H, int W, int C,
                                           int scale, int out_nchw, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(x && y && scale >= 1 && B > 0 && H > 0 && W > 0 && C > 0, "upsample: bad arguments");
  return upsample_launch(x, ldx, y, ldy, B, H, W, C, scale, out_nchw, nullptr, nullptr, nullptr, stream);
}

/* CoCUpsample (coc_fpn_dual.py:15-26: 1x1 conv -> BatchNorm -> ReLU -> bilinear) without the low-resolution activation: the
 * upsample gathers ReLU(A (z - S) + D) from the conv output z (round 5, K11).  Same bits as vrnet_affine_f32 (pre = 1) followed
 * by vrnet_upsample_bilinear_f32. */
extern "C" int vrnet_bn_relu_upsample_bilinear_f32(const float* z, long ldz, const float* A, const float* D, const float* S,
                                                   float* y, long ldy, int B, int H, int W, int C, int scale, int out_nchw,
                                                   void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(z && A && D && S && y && scale >= 1 && B > 0 && H > 0 && W > 0 && C > 0, "bn_relu_upsample: bad arguments");
  return upsample_launch(z, ldz, y, ldy, B, H, W, C, scale, out_nchw, A, D, S, stream);
}

extern "C" int vrnet_upsample_bilinear_bwd_f32(const float* dy, long lddy, int dy_nchw, float* dx, long lddx, int B,
                                               int H, int W, int C, int scale, int accumulate, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(dy && dx && scale >= 1, "upsample_bwd: bad arguments");
  if (!dy_nchw && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && vr_aligned16(dy) && vr_aligned16(dx))
    hipLaunchKernelGGL(upsample_bwd_vec_kernel, dim3(grid_for((long)B * H * W * (C / 4), 256)), dim3(256), 0, vr_stream(stream),
                       dy, lddy, dx, lddx, B, H, W, C, H * scale, W * scale, accumulate);
  else
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((long)B * H * W * C, 256)), dim3(256), 0, vr_stream(stream), dy,
                       lddy, dy_nchw, dx, lddx, B, H, W, C, H * scale, W * scale, accumulate);
  VR_LAUNCH_CHECK("upsample_bwd");
  return VR_OK;
}

extern "C" long vrnet_reduce_workspace(void) { return 8192L * 4 * 8 + 256; }

extern "C" int vrnet_minmax_f32(const float* p, long n, float* mm, void* workspace, long workspace_bytes, void* stream) {
  VR_CHECK_ARG(p && mm && workspace && n > 0, "minmax: bad arguments");
  VR_CHECK_ARG(workspace_bytes >= vrnet_reduce_workspace(), "minmax: workspace too small");
  const int nb = (int)grid_for(n, 2048, 2048);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, vr_stream(stream), p, n, partial);
  VR_LAUNCH_CHECK("minmax_partial");
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, vr_stream(stream), partial, nb, mm);
  VR_LAUNCH_CHECK("minmax_final");
  return VR_OK;
}

extern "C" int vrnet_enhance_mul_f32(const float* p, const float* x, const float* mm, float* out, long n, void* stream) {
  VR_CHECK_ARG(p && x && mm && out && n > 0, "enhance_mul: bad arguments");
  hipLaunchKernelGGL(enhance_mul_kernel, dim3(grid_for(n)), dim3(256), 0, vr_stream(stream), p, x, mm, out, n);
  VR_LAUNCH_CHECK("enhance_mul");
  return VR_OK;
}

/* t = (1 + data_normal(p)) * x with the batch-wide (min, max) of p found on the way (left in mm[0..1] for the backward pass):
 * vrnet_minmax_f32 + vrnet_enhance_mul_f32 as TWO launches instead of three (the final min / max step runs inside the apply
 * kernel).  Same results as the two calls. */
extern "C" int vrnet_enhance_fwd_f32(const float* p, const float* x, float* mm, float* out, long n, void* workspace,
                                     long workspace_bytes, void* stream) {
  VR_CHECK_ARG(p && x && mm && out && workspace && n > 0, "enhance_fwd: bad arguments");
  VR_CHECK_ARG(workspace_bytes >= vrnet_reduce_workspace(), "enhance_fwd: workspace too small");
  const int nb = (int)grid_for(n, 2048, 512);
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t st = vr_stream(stream);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, st, p, n, partial);
  VR_LAUNCH_CHECK("minmax_partial");
  const int vec = ((n & 3) == 0 && vr_aligned16(p) && vr_aligned16(x) && vr_aligned16(out)) ? 1 : 0;
  hipLaunchKernelGGL(enhance_fwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, st, p, x, partial, nb, mm, out, n, vec);
  VR_LAUNCH_CHECK("enhance_fwd");
  return VR_OK;
}

extern "C" int vrnet_enhance_bwd_f32(const float* dt, const float* x, const float* p, const float* mm, float* dx,
                                     float* dp, long n, int accumulate_dx, void* workspace, long workspace_bytes,
                                     void* stream) {
  VR_CHECK_ARG(dt && x && p && mm && dx && dp && workspace && n > 0, "enhance_bwd: bad arguments");
  VR_CHECK_ARG(workspace_bytes >= vrnet_reduce_workspace(), "enhance_bwd: workspace too small");
  const int nb = (int)grid_for(n, 2048, 256);       // few partials: every workgroup of the apply kernel re-adds them
  double* partial = reinterpret_cast<double*>(workspace);
  hipStream_t st = vr_stream(stream);
  hipLaunchKernelGGL(enhance_bwd_partial_kernel, dim3(nb), dim3(256), 0, st, dt, x, p, mm, n, partial);
  VR_LAUNCH_CHECK("enhance_bwd_partial");
  hipLaunchKernelGGL(enhance_bwd_apply_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, st, dt, x, p, mm, partial, nb, dx, dp, n,
                     accumulate_dx);
  VR_LAUNCH_CHECK("enhance_bwd_apply");
  return VR_OK;
}

extern "C" int vrnet_sa_coef_fwd(const double* mom, const float* cw, const float* cb, const float* sw, const float* sb,
                                 const float* gnw, const float* gnb, int B, long HW, int C, int G, float* P, float* Q,
                                 float* Mn, void* stream) {
  VR_CHECK_ARG(mom && cw && cb && sw && sb && gnw && gnb && P && Q && Mn, "sa_coef_fwd: null tensor");
  VR_CHECK_ARG(G > 0 && C % (2 * G) == 0 && C / (2 * G) > 0, "sa_coef_fwd: channels %d not divisible by 2*G=%d", C, 2 * G);
  SaParams sp{cw, cb, sw, sb, gnw, gnb};
  hipLaunchKernelGGL(sa_coef_fwd_kernel, dim3(vr_cdiv((long)B * C, 256)), dim3(256), 0, vr_stream(stream), mom, sp, B, HW,
                     C, G, P, Q, Mn);
  VR_LAUNCH_CHECK("sa_coef_fwd");
  return VR_OK;
}

extern "C" int vrnet_sa_apply_f32(const float* x, long ldx, const float* P, const float* Q, const float* Mn, float* y,
                                  long ldy, int B, long HW, int C, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(x && P && Q && Mn && y && C % 2 == 0, "sa_apply: bad arguments");
  hipLaunchKernelGGL(sa_apply_kernel, dim3(grid_for(HW * C), B), dim3(256), 0, vr_stream(stream), x, ldx, P, Q, Mn, y, ldy,
                     HW, C);
  VR_LAUNCH_CHECK("sa_apply");
  return VR_OK;
}

extern "C" long vrnet_sa_bwd_workspace(int B, long HW, int C) {
  int TPR, ncb, nchunks;
  long rows;
  sa_plan(HW, C, &TPR, &ncb, &nchunks, &rows);
  return ((long)B * nchunks * C * 2 + (long)B * C * 2) * 8 + 256;
}

extern "C" int vrnet_sa_bwd_f32(const float* dy, long lddy, const float* x, long ldx, const float* P, const float* Q,
                                const float* Mn, const double* mom, const float* cw, const float* cb, const float* sw, const float* sb,
                                const float* gnw, const float* gnb, float* dx, long lddx, float* dcw, float* dcb,
                                float* dsw, float* dsb, float* dgnw, float* dgnb, float* EF /*[2][B][C] scratch*/, int B,
                                long HW, int C, int G, int accumulate_dx, int accumulate_params, void* workspace,
                                long workspace_bytes, void* stream) {
  if (vr_ablated("misc")) return VR_OK;
  VR_CHECK_ARG(dy && x && P && Q && Mn && mom && dx && EF && workspace, "sa_bwd: null tensor");
  VR_CHECK_ARG(workspace_bytes >= vrnet_sa_bwd_workspace(B, HW, C), "sa_bwd: workspace too small");
  int TPR, ncb, nchunks;
  long rows;
  sa_plan(HW, C, &TPR, &ncb, &nchunks, &rows);
  double* partial = reinterpret_cast<double*>(workspace);
  double* T = partial + (long)B * nchunks * C * 2;
  hipStream_t st = vr_stream(stream);
  hipLaunchKernelGGL(sa_bwd_reduce_kernel, dim3(nchunks, B, ncb), dim3(256), 256 * 2 * sizeof(double), st, dy, lddy, x,
                     ldx, P, Q, Mn, HW, C, TPR, rows, nchunks, partial);
  VR_LAUNCH_CHECK("sa_bwd_reduce");
  hipLaunchKernelGGL(sa_reduce_chunks_kernel, dim3(vr_cdiv((long)B * C * 2, 16)), dim3(256), 0, st, partial, T, B,
                     nchunks, C);
  VR_LAUNCH_CHECK("sa_reduce_chunks");
  SaParams sp{cw, cb, sw, sb, gnw, gnb};
  const int nb1 = (int)vr_cdiv((long)B * C, 256);
  float* E = EF;
  float* F = EF + (long)B * C;
  hipLaunchKernelGGL(sa_coef_bwd_kernel, dim3(nb1 + C / (2 * G)), dim3(256), 0, st, T, mom, sp, B, HW, C, G, nb1, E, F, dcw, dcb,
                     dsw, dsb, dgnw, dgnb, accumulate_params);
  VR_LAUNCH_CHECK("sa_coef_bwd");
  hipLaunchKernelGGL(sa_bwd_apply_kernel, dim3(grid_for(HW * C), B), dim3(256), 0, st, dy, lddy, x, ldx, P, Q, Mn, E, F, dx,
                     lddx, HW, C, accumulate_dx);
  VR_LAUNCH_CHECK("sa_bwd_apply");
  return VR_OK;
}

extern "C" int vrnet_patch_gather_f32(const float* x, long ldx, const float* pos, float* out, int B, int H, int W, int C,
                                      int CP, int k, void* stream) {
  VR_CHECK_ARG(x && out && (CP == 0 || pos) && B > 0 && C > 0 && CP >= 0 && k > 0 && H % k == 0 && W % k == 0 && ldx >= C,
               "patch_gather: bad arguments");
  hipLaunchKernelGGL(patch_gather_kernel, dim3(grid_for((long)B * H * W * (C + CP))), dim3(256), 0, vr_stream(stream), x,
                     ldx, pos, out, B, H, W, C, CP, k);
  VR_LAUNCH_CHECK("patch_gather");
  return VR_OK;
}

extern "C" int vrnet_patch_scatter_f32(const float* dp, float* dx, long lddx, int B, int H, int W, int C, int CP, int k,
                                       int accumulate, void* stream) {
  VR_CHECK_ARG(dp && dx && B > 0 && C > 0 && CP >= 0 && k > 0 && H % k == 0 && W % k == 0 && lddx >= C,
               "patch_scatter: bad arguments");
  hipLaunchKernelGGL(patch_scatter_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0, vr_stream(stream), dp, dx,
                     lddx, B, H, W, C, CP, k, accumulate);
  VR_LAUNCH_CHECK("patch_scatter");
  return VR_OK;
}

extern "C" int vrnet_weight_ohwi_f32(const float* src, float* dst, int Cout, int Cin, int kh, int kw, int dir,
                                     int accumulate, void* stream) {
  VR_CHECK_ARG(src && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0 && (dir == 0 || dir == 1), "weight_ohwi: bad arguments");
  const long total = (long)Cout * Cin * kh * kw;
  hipLaunchKernelGGL(ohwi_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, vr_stream(stream), src, dst, Cout, Cin, kh * kw,
                     dir, accumulate);
  VR_LAUNCH_CHECK("weight_ohwi");
  return VR_OK;
}
