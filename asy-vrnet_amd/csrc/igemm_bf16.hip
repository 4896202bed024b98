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
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmArgs p) {
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

extern "C" int vrnet_pack_weight_t_f32(const float* w_oihw, const float* kscale, float* w_tcn, int Cout, int Cin, int kh,
                                       int kw, void* stream) {
  VR_CHECK_ARG(w_oihw && w_tcn && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "pack_weight_t: bad arguments");
  const long total = (long)kh * kw * Cout * Cin;
  hipLaunchKernelGGL(pack_weight_t_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, vr_stream(stream), w_oihw, kscale, w_tcn,
                     Cout, Cin, kh * kw);
  VR_LAUNCH_CHECK("pack_weight_t");
  return VR_OK;
}
