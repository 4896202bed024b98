// Direct kernels for 1x1 convolutions with a NARROW output (Cout <= 16) over wide inputs: the prediction convs of the
// head (256 -> 4 / 1 / num_classes, head/decouplehead.py:36-40,74-86) and the seg-logit conv (128 -> 9,
// neck/coc_fpn_dual.py:19-22).  On an MFMA tile their N is padded to 32 (> 70 % idle columns) and the weight gradient
// contracts 32 768 rows into a 4 x 256 matrix through 128 x 32 tiles: 0.1 - 3 TFLOP/s, 15 - 130 us per launch in the
// round-2 profile.  These are HBM streams -- every input row is read exactly once -- so they are written as such:
// a row of K channels is covered by QP = K / 4 threads holding one float4 each (whole 128-byte lines per 8 lanes), RL =
// 256 / QP rows are in flight per workgroup pass.
// All arithmetic is written per scalar component: a "scalar x float4" product compiles to packed-fp32 instructions with
// op_sel operand broadcast, and such sequences gave wrong sums in a few per cent of launches whenever another kernel
// shared the CU (round 2: the conv epilogue statistics; round 3: the first version of these kernels -- single components
// of single rows of a weight gradient off by 1e-3, only beside concurrent streams, never alone).
#include "common.h"

namespace narrow {

constexpr int NMAX = 16;

struct Args {
  const float* a; long lda;        // forward: x [M][K]; data gradient: dy [M][N]
  const float* w;                  // [N][K] (the OIHW tensor of a 1x1 conv)
  const float* bias;               // forward only
  float* y; long ldy;              // forward: [M][N] or NCHW; data gradient: dx [M][K]
  long M; int K, N;
  int QP, RL, qshift;              // threads per row (power of two >= K / 4), rows per pass, log2(QP)
  int out_nchw, out_ctot, out_coff; long HW;
  int accumulate;
};

// y[m][n] = bias[n] + sum_k x[m][k] w[n][k].  Thread (q, r): quad q of row r; its N x 4 weights stay in registers; the QP
// partial dot products of a row are added by a butterfly inside the wave (QP <= 64).
template <int N>
__global__ __launch_bounds__(256) void fwd_kernel(const Args p) {
  const int tid = threadIdx.x, q = tid & (p.QP - 1), r = tid >> p.qshift;
  const bool qlive = 4 * q < p.K;
  f32x4 w4[N];
#pragma unroll
  for (int n = 0; n < N; ++n) w4[n] = qlive ? *reinterpret_cast<const f32x4*>(p.w + (long)n * p.K + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (long m0 = (long)blockIdx.x * p.RL; m0 < p.M; m0 += (long)gridDim.x * p.RL) {
    const long m = m0 + r;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (m < p.M && qlive) x = *reinterpret_cast<const f32x4*>(p.a + m * p.lda + 4 * q);
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = (x[0] * w4[n][0] + x[1] * w4[n][1]) + (x[2] * w4[n][2] + x[3] * w4[n][3]);
    for (int o = p.QP >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += __shfl_xor(acc[n], o, 64);
    }
    if (q == 0 && m < p.M) {
      if (p.out_nchw) {
        const long b = m / p.HW, pix = m - b * p.HW;
#pragma unroll
        for (int n = 0; n < N; ++n) {
          float* d = p.y + ((b * p.out_ctot + p.out_coff + n) * p.HW + pix);
          const float v = acc[n] + (p.bias ? p.bias[n] : 0.f);
          *d = p.accumulate ? *d + v : v;
        }
      } else {
#pragma unroll
        for (int n = 0; n < N; ++n) {
          float* d = p.y + m * p.ldy + n;
          const float v = acc[n] + (p.bias ? p.bias[n] : 0.f);
          *d = p.accumulate ? *d + v : v;
        }
      }
    }
  }
}

// dx[m][k] (+)= sum_n dy[m][n] w[n][k]
template <int N>
__global__ __launch_bounds__(256) void dgrad_kernel(const Args p) {
  const int tid = threadIdx.x, q = tid & (p.QP - 1), r = tid >> p.qshift;
  const bool qlive = 4 * q < p.K;
  f32x4 w4[N];
#pragma unroll
  for (int n = 0; n < N; ++n) w4[n] = qlive ? *reinterpret_cast<const f32x4*>(p.w + (long)n * p.K + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (long m0 = (long)blockIdx.x * p.RL; m0 < p.M; m0 += (long)gridDim.x * p.RL) {
    const long m = m0 + r;
    if (m >= p.M || !qlive) continue;
    const float* g = p.a + m * p.lda;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float gn = g[n];
      v0 = fmaf(gn, w4[n][0], v0);
      v1 = fmaf(gn, w4[n][1], v1);
      v2 = fmaf(gn, w4[n][2], v2);
      v3 = fmaf(gn, w4[n][3], v3);
    }
    f32x4* d = reinterpret_cast<f32x4*>(p.y + m * p.ldy + 4 * q);
    if (p.accumulate) {
      const f32x4 old = *d;
      v0 += old[0]; v1 += old[1]; v2 += old[2]; v3 += old[3];
    }
    *d = f32x4{v0, v1, v2, v3};
  }
}

// Weight gradient: slab[split][n][k] = sum over the split's rows of dy[m][n] x[m][k]; bslab[split][n] = sum dy[m][n].
// Thread (q, r) adds the rows r, r + RL, ... of its split; the RL partial sums are added in a fixed order through LDS.
template <int N>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* x, long ldx, const float* dy, long lddy, long M, int K, int QP,
                                                    int RL, int qshift, long rows_per_split, float* slab, float* bslab) {
  extern __shared__ float red[];       // [RL][N][4 QP]
  const int tid = threadIdx.x, q = tid & (QP - 1), r = tid >> qshift;
  const bool qlive = 4 * q < K;
  const long m_begin = (long)blockIdx.x * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  float acc[N][4];
  float bsum[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; bsum[n] = 0.f; }
  long m = m_begin + r;
  for (; m + RL < m_end; m += 2L * RL) {         // two rows per trip: both loads in flight before the FMAs
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
    if (qlive) {
      x0 = *reinterpret_cast<const f32x4*>(x + m * ldx + 4 * q);
      x1 = *reinterpret_cast<const f32x4*>(x + (m + RL) * ldx + 4 * q);
    }
    const float* g0 = dy + m * lddy;
    const float* g1 = dy + (m + RL) * lddy;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float a = g0[n], b = g1[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n][j] = fmaf(b, x1[j], fmaf(a, x0[j], acc[n][j]));
      bsum[n] += a + b;
    }
  }
  for (; m < m_end; m += RL) {
    const f32x4 x0 = qlive ? *reinterpret_cast<const f32x4*>(x + m * ldx + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* g0 = dy + m * lddy;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const float a = g0[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n][j] = fmaf(a, x0[j], acc[n][j]);
      bsum[n] += a;
    }
  }
  const int W4 = 4 * QP;
#pragma unroll
  for (int n = 0; n < N; ++n)
    *reinterpret_cast<f32x4*>(red + ((long)r * N + n) * W4 + 4 * q) = f32x4{acc[n][0], acc[n][1], acc[n][2], acc[n][3]};
  __syncthreads();
  if (r == 0 && qlive) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float s0 = acc[n][0], s1 = acc[n][1], s2 = acc[n][2], s3 = acc[n][3];
      for (int k = 1; k < RL; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(red + ((long)k * N + n) * W4 + 4 * q);
        s0 += t[0]; s1 += t[1]; s2 += t[2]; s3 += t[3];
      }
      *reinterpret_cast<f32x4*>(slab + ((long)blockIdx.x * N + n) * K + 4 * q) = f32x4{s0, s1, s2, s3};
    }
  }
  if (bslab) {        // (every thread of a row group holds the same row sums: quad 0 of each group reports them)
    __syncthreads();
    if (q == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) red[r * N + n] = bsum[n];
    }
    __syncthreads();
    if (tid < N) {
      float s = 0.f;
      for (int k = 0; k < RL; ++k) s += red[k * N + tid];
      bslab[(long)blockIdx.x * N + tid] = s;
    }
  }
}

static void plan(int K, int* QP, int* RL, int* qshift) {
  int qp = 1, sh = 0;
  while (qp < K / 4) { qp <<= 1; ++sh; }
  *QP = qp; *RL = 256 / qp; *qshift = sh;
}

}  // namespace narrow

// Shapes: 1x1, stride 1, K % 4 == 0, 16-byte rows.  forward / data gradient: K <= 256 (a row's threads share a wave) and
// at most 12 output channels (the prediction / logit convs have 1, 4, num_classes, num_seg_classes = 9; a 16-channel
// layer -- phi = nano's stage-0 width -- fills half an MFMA column block and stays on the MFMA kernels).
bool vr_narrow_conv_ok(int Cin, int Cout, int kh, int kw, int stride, int pad) {
  return kh == 1 && kw == 1 && stride == 1 && pad == 0 && Cout <= 12 && Cin % 4 == 0 && Cin >= 16 && Cin <= 256;
}
bool vr_narrow_wgrad_ok(int Cin, int Cout, int kh, int kw, int stride, int pad) {
  return kh == 1 && kw == 1 && stride == 1 && pad == 0 && Cout <= narrow::NMAX && Cin % 4 == 0 && Cin >= 16 && Cin <= 1024;
}

#define VR_NARROW_SWITCH(N_, CALL)                                                                                   \
  switch (N_) {                                                                                                      \
    case 1: { constexpr int NN = 1; CALL; } break;                                                                   \
    case 2: { constexpr int NN = 2; CALL; } break;                                                                   \
    case 3: { constexpr int NN = 3; CALL; } break;                                                                   \
    case 4: { constexpr int NN = 4; CALL; } break;                                                                   \
    case 5: { constexpr int NN = 5; CALL; } break;                                                                   \
    case 6: { constexpr int NN = 6; CALL; } break;                                                                   \
    case 7: { constexpr int NN = 7; CALL; } break;                                                                   \
    case 8: { constexpr int NN = 8; CALL; } break;                                                                   \
    case 9: { constexpr int NN = 9; CALL; } break;                                                                   \
    case 10: { constexpr int NN = 10; CALL; } break;                                                                 \
    case 11: { constexpr int NN = 11; CALL; } break;                                                                 \
    case 12: { constexpr int NN = 12; CALL; } break;                                                                 \
    case 13: { constexpr int NN = 13; CALL; } break;                                                                 \
    case 14: { constexpr int NN = 14; CALL; } break;                                                                 \
    case 15: { constexpr int NN = 15; CALL; } break;                                                                 \
    default: { constexpr int NN = 16; CALL; } break;                                                                 \
  }

// mode 0: y = conv(a); mode 1: y (B*HW x Cin) (+)= d/dx given a = dy (row stride lda)
int vr_narrow_conv(int mode, const float* a, long lda, const float* w, const float* bias, float* y, long ldy, long M, long HW,
                   int Cin, int Cout, int out_nchw, int out_ctot, int out_coff, int accumulate, hipStream_t st) {
  narrow::Args p{};
  p.a = a; p.lda = lda; p.w = w; p.bias = bias; p.y = y; p.ldy = ldy; p.M = M; p.K = Cin; p.N = Cout;
  narrow::plan(Cin, &p.QP, &p.RL, &p.qshift);
  p.out_nchw = out_nchw; p.out_ctot = out_ctot; p.out_coff = out_coff; p.HW = HW; p.accumulate = accumulate;
  long blocks = vr_cdiv(M, p.RL * 8L);            // >= 8 passes per workgroup (the weights are loaded once per workgroup)
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), block(256);
  if (mode == 0) {
    VR_NARROW_SWITCH(Cout, hipLaunchKernelGGL((narrow::fwd_kernel<NN>), grid, block, 0, st, p));
  } else {
    VR_NARROW_SWITCH(Cout, hipLaunchKernelGGL((narrow::dgrad_kernel<NN>), grid, block, 0, st, p));
  }
  VR_LAUNCH_CHECK("conv2d(narrow)");
  return VR_OK;
}

// Row splits of the narrow weight gradient: <= 1024 slabs of >= 32 rows (a 16 x 16 map at bs 8 still yields 64 workgroups)
static void narrow_wgrad_plan(long M, long* rows, int* S) {
  long r = vr_cdiv(M, 1024);
  if (r < 32) r = 32;
  *rows = r;
  *S = (int)vr_cdiv(M, r);
}
long vr_narrow_wgrad_workspace(long M, int Cin, int Cout) {
  long rows; int S;
  narrow_wgrad_plan(M, &rows, &S);
  return (long)S * ((long)Cout * Cin + Cout) * 4 + 256;
}
// Fills slab [S][Cout][Cin] and (want_bias) bslab [S][Cout] inside `workspace`; returns their addresses and S.
int vr_narrow_wgrad(const float* x, long ldx, const float* dy, long lddy, long M, int Cin, int Cout, void* workspace,
                    int want_bias, float** slab_out, float** bslab_out, int* splits, hipStream_t st) {
  long rows; int S;
  narrow_wgrad_plan(M, &rows, &S);
  float* slab = reinterpret_cast<float*>(workspace);
  float* bslab = want_bias ? slab + (long)S * Cout * Cin : nullptr;
  int QP, RL, qshift;
  narrow::plan(Cin, &QP, &RL, &qshift);
  const size_t lds = (size_t)RL * Cout * 4 * QP * sizeof(float);
  VR_NARROW_SWITCH(Cout, hipLaunchKernelGGL((narrow::wgrad_kernel<NN>), dim3(S), dim3(256), lds, st, x, ldx, dy, lddy, M, Cin, QP, RL,
                                            qshift, rows, slab, bslab));
  VR_LAUNCH_CHECK("conv2d_wgrad(narrow)");
  *slab_out = slab; *bslab_out = bslab; *splits = S;
  return VR_OK;
}
