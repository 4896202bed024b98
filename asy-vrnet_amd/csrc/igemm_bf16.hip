// bf16-operand variant of the implicit-GEMM convolution (BASELINE configs[2..4]: "bf16 with MFMA conv path"):
// activations and weights stay fp32 in HBM, are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while
// they are staged into LDS, multiplied on v_mfma_f32_32x32x16_bf16 and accumulated in fp32; the epilogue is the
// fp32 one.  8x the contraction depth per MFMA at half its cycles: the dense layers become HBM / L2 streams.
//   C[m, n] = sum_{t, k} A[src(m, t), k] * Bt[t][n][k]      -- BOTH operands contraction-contiguous:
//   mode 0 (forward)        Bt = packed weights [t][Cout][Cin]                       (vrnet_pack_weight_f32)
//   mode 1 (data gradient)  Bt = TRANSPOSED pack [t][Cin][Cout] (x kscale[Cout])    (vrnet_pack_weight_t_f32)
// 64 x 64 x 64 tile, 4 waves, one 32x32 accumulator each; operand tiles global -> registers (8 float4 per thread in
// flight) -> bf16 rows of 72 elements (144-byte stride: every ds_read_b128 fragment read is conflict-free).
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmArgs p_in) {
  const IgemmArgs p = igemm_select_stream(p_in, blockIdx.x * 64);
  constexpr int BM = 64, BN = 64, BK = 64, LD = BK + 8;
  constexpr int OPER_BYTES = 2 * BM * LD * 2, EPI_BYTES = 4 * 32 * STAGE_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES];
  __shared__ unsigned tapmask_s;
  __shared__ unsigned char taps_s[32];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Bs = As + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int nkb = (p.CK + BK - 1) / BK;
  const int T = p.kh * p.kw;

  // loader roles: 4 float4 of A and 4 of Bt per thread and step; slot s = tid + 256 i -> (row s / 16, quad s % 16)
  int a_b[4], a_y[4], a_x[4];
  bool a_ok[4];
  const int q4 = 4 * (tid & 15);                 // first contraction channel of this thread's quads
  const int r0 = tid >> 4;                       // rows r0, r0 + 16, r0 + 32, r0 + 48
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 16 * i;
    a_ok[i] = m < p.M;
    igemm_row_to_pixel(p, a_ok[i] ? m : 0, a_b[i], a_y[i], a_x[i]);
  }
  auto src_of = [&](int i, int ky, int kx, int& sy, int& sx) -> bool {
    bool ok = a_ok[i];
    if (MODE == 0) {
      sy = a_y[i] * p.stride - p.pad + ky * p.dil;
      sx = a_x[i] * p.stride - p.pad + kx * p.dil;
    } else {
      const int ty = a_y[i] + p.pad - ky * p.dil, tx = a_x[i] + p.pad - kx * p.dil;
      ok = ok && ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0;
      sy = ty / p.stride;
      sx = tx / p.stride;
    }
    return ok && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;
  };
  int ntaps = T;
  const bool use_list = T > 1 && T <= 32;
  if (use_list) {
    if (tid == 0) tapmask_s = 0u;
    __syncthreads();
    unsigned mine = 0u;
    for (int t = 0; t < T; ++t) {
      const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int sy, sx;
        if (src_of(i, ky, kx, sy, sx)) mine |= 1u << t;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o, 64);
    if (lane == 0 && mine) atomicOr(&tapmask_s, mine);
    __syncthreads();
    if (tid == 0) {
      const unsigned mk = tapmask_s;
      int c = 0;
      for (int t = 0; t < T; ++t)
        if (mk & (1u << t)) taps_s[c++] = (unsigned char)t;
    }
    __syncthreads();
    ntaps = __popc(tapmask_s);
  }
  const int nsteps = ntaps * nkb;

  const float* a_ptr[4];
  const float* b_ptr[4];
  auto setup_tap = [&](int t) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sy, sx;
      const bool ok = src_of(i, ky, kx, sy, sx);
      a_ptr[i] = ok ? p.a + ((long)(a_b[i] * p.SH + sy) * p.SW + sx) * p.lda + q4 : nullptr;
      const int n = n0 + r0 + 16 * i;
      b_ptr[i] = n < p.CN ? p.w + (long)t * p.wtap + (long)n * p.CK + q4 : nullptr;
    }
  };
  f32x4 areg[4], breg[4];
  int ld_ti = 0, ld_kb = 0;
  auto load_tiles = [&]() {
    if (ld_kb == 0) setup_tap(use_list ? (int)taps_s[ld_ti] : ld_ti);
    const int kc = ld_kb * BK + q4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      areg[i] = (a_ptr[i] != nullptr && kc < p.CK) ? *reinterpret_cast<const f32x4*>(a_ptr[i] + ld_kb * BK) : zero;
      breg[i] = (b_ptr[i] != nullptr && kc < p.CK) ? *reinterpret_cast<const f32x4*>(b_ptr[i] + ld_kb * BK) : zero;
    }
    if (++ld_kb == nkb) {
      ld_kb = 0;
      ++ld_ti;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x4 pa = {(__bf16)areg[i][0], (__bf16)areg[i][1], (__bf16)areg[i][2], (__bf16)areg[i][3]};
      const bf16x4 pb = {(__bf16)breg[i][0], (__bf16)breg[i][1], (__bf16)breg[i][2], (__bf16)breg[i][3]};
      *reinterpret_cast<bf16x4*>(As + (r0 + 16 * i) * LD + q4) = pa;
      *reinterpret_cast<bf16x4*>(Bs + (r0 + 16 * i) * LD + q4) = pb;
    }
  };

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int h = lane >> 5;
  const __bf16* a_frag = As + (wm * 32 + (lane & 31)) * LD + 8 * h;
  const __bf16* b_frag = Bs + (wn * 32 + (lane & 31)) * LD + 8 * h;

  if (nsteps > 0) load_tiles();
  for (int s = 0; s < nsteps; ++s) {
    store_tiles();
    __syncthreads();
    if (s + 1 < nsteps) load_tiles();
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a_frag + 16 * kk);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b_frag + 16 * kk);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0][0], 0, 0, 0);
    }
    __syncthreads();
  }
  igemm_epilogue<1, 1, 2, 2>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0);
}

// Weight gradient on the bf16 MFMA: dW[t][n][c] = sum_m dy[m, n] * x[src(m, t), c].  The contraction index is the ROW
// of both operands, so the fragments (8 consecutive m of one column) are columns of the staged tiles: dy and x tiles
// are stored row-major ([m][64 channels], 144-byte rows, coalesced b64 stores) and read with gfx950's transposing
// ds_read_b64_tr_b16 (per 16-lane group a 4-row x 16-column block arrives column-major: lane i gets column i of the
// 4 rows) -- no transposing stores, no shuffles.  Same slabs / reduce pass as the fp32 weight gradient.
template <bool IDENT>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgradArgs p) {
  constexpr int BKM = 64, LD = 72;
  __shared__ __attribute__((aligned(16))) __bf16 Ys[BKM * LD];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[BKM * LD];
  __shared__ float bred[16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x;
  const int ct = bid % p.c_tiles; bid /= p.c_tiles;
  const int nt = bid % p.n_tiles; bid /= p.n_tiles;
  const int t = bid;
  const int ky = t / p.kw, kx = t - ky * p.kw;
  const int n0 = nt * 64, c0 = ct * 64;
  int m_begin, m_end, split;
  wgrad_rows(p, m_begin, m_end, split);
  const bool do_bias = p.bslab != nullptr && ct == 0 && t == 0;
  const int q4 = 4 * (tid & 15), r0 = tid >> 4;          // this thread's channel quad, rows r0 + 16 i of a step
  f32x4 yreg[4], xreg[4];
  f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
  auto load_tiles = [&](int mb) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + r0 + 16 * i;
      yreg[i] = (m < m_end && n0 + q4 < p.Cout) ? *reinterpret_cast<const f32x4*>(p.dy + (long)m * p.lddy + n0 + q4) : zero;
      f32x4 v = zero;
      if (m < m_end && c0 + q4 < p.Cin) {
        if (IDENT) {
          v = *reinterpret_cast<const f32x4*>(p.x + (long)m * p.ldx + c0 + q4);
        } else {
          const int ox = m % p.OW;
          const int q = m / p.OW;
          const int oy = q % p.OH, b = q / p.OH;
          const int sy = oy * p.stride - p.pad + ky * p.dil, sx = ox * p.stride - p.pad + kx * p.dil;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + ((long)(b * p.H + sy) * p.W + sx) * p.ldx + c0 + q4);
        }
      }
      xreg[i] = v;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (do_bias) bacc += yreg[i];
      const bf16x4 py = {(__bf16)yreg[i][0], (__bf16)yreg[i][1], (__bf16)yreg[i][2], (__bf16)yreg[i][3]};
      const bf16x4 px = {(__bf16)xreg[i][0], (__bf16)xreg[i][1], (__bf16)xreg[i][2], (__bf16)xreg[i][3]};
      *reinterpret_cast<bf16x4*>(Ys + (r0 + 16 * i) * LD + q4) = py;
      *reinterpret_cast<bf16x4*>(Xs + (r0 + 16 * i) * LD + q4) = px;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // transposed-read addresses: lane 4q + pp of its 16-lane group g supplies row (8 h + q) [+ 4], columns 16 g + 4 pp
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const __bf16* y_tr = Ys + (8 * h + q) * LD + wm * 32 + 16 * g + 4 * pp;
  const __bf16* x_tr = Xs + (8 * h + q) * LD + wn * 32 + 16 * g + 4 * pp;
  typedef __attribute__((address_space(3))) bf16x4* lds4;

  if (m_begin < m_end) load_tiles(m_begin);
  for (int mb = m_begin; mb < m_end; mb += BKM) {
    store_tiles();
    __syncthreads();
    if (mb + BKM < m_end) load_tiles(mb + BKM);
#pragma unroll
    for (int kk = 0; kk < BKM / 16; ++kk) {
      const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(y_tr + (16 * kk) * LD));
      const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(y_tr + (16 * kk + 4) * LD));
      const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(x_tr + (16 * kk) * LD));
      const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(x_tr + (16 * kk + 4) * LD));
      const bf16x8 fa = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const bf16x8 fb = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const long T = (long)p.kh * p.kw;
  float* slab = p.slab + ((long)split * T + t) * p.Cout * p.Cin;
  const int c = c0 + wn * 32 + (lane & 31);
  if (c < p.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < p.Cout) slab[(long)n * p.Cin + c] = acc[r];
    }
  }
  if (do_bias) {      // fp32 column sums of dy: 16 row groups x 64 columns through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) bred[r0][q4 + j] = bacc[j];
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Cout) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += bred[r][tid];
      p.bslab[(long)split * p.Cout + n0 + tid] = sum;
    }
  }
}

// [t][c][n] = w_oihw[n][c][t] * kscale[n]: the data gradient's Bt (rows = input channels, contraction over Cout)
__global__ void pack_weight_t_kernel(const float* w, const float* kscale, float* out, int Cout, int Cin, int T) {
  const long total = (long)T * Cout * Cin;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int n = e % Cout;
  const long q = e / Cout;
  const int c = q % Cin;
  const int t = q / Cin;
  out[e] = w[((long)n * Cin + c) * T + t] * (kscale ? kscale[n] : 1.f);
}

}  // namespace

// internal entry used by vrnet_conv2d_f32 (igemm.hip) when precision == 1
int vr_igemm_bf16_launch(const void* args, int mode, hipStream_t st) {
  const IgemmArgs& p = *reinterpret_cast<const IgemmArgs*>(args);
  dim3 grid(vr_cdiv(p.M, 64), vr_cdiv(p.CN, 64)), block(256);
  if (mode == 0) hipLaunchKernelGGL((igemm_bf16_kernel<0>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((igemm_bf16_kernel<1>), grid, block, 0, st, p);
  return VR_OK;
}

// internal entry used by vrnet_conv2d_wgrad_f32 (igemm.hip) when precision == 1
int vr_wgrad_bf16_launch(const void* args, int ident, int blocks_x, int splits, int streams, hipStream_t st) {
  const WgradArgs& p = *reinterpret_cast<const WgradArgs*>(args);
  dim3 grid(blocks_x, splits, streams), block(256);
  if (ident) hipLaunchKernelGGL((wgrad_bf16_kernel<true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((wgrad_bf16_kernel<false>), grid, block, 0, st, p);
  return VR_OK;
}

extern "C" int vrnet_pack_weight_t_f32(const float* w_oihw, const float* kscale, float* w_tcn, int Cout, int Cin, int kh,
                                       int kw, void* stream) {
  VR_CHECK_ARG(w_oihw && w_tcn && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "pack_weight_t: bad arguments");
  const long total = (long)kh * kw * Cout * Cin;
  hipLaunchKernelGGL(pack_weight_t_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, vr_stream(stream), w_oihw, kscale, w_tcn,
                     Cout, Cin, kh * kw);
  VR_LAUNCH_CHECK("pack_weight_t");
  return VR_OK;
}
