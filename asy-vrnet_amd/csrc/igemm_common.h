// Shared by the implicit-GEMM kernels (igemm.hip: fp32 MFMA; igemm_bf16.hip: bf16 MFMA): argument block, GEMM-row ->
// pixel maps and the fused epilogue.
#pragma once
#include "common.h"

extern "C" {
// include/vrnet_hip.h: optional column statistics of a conv's stored outputs
typedef struct vrnet_conv_colstats {
  double* partial;              // [ceil(M/32)][N][2]: per output column, (sum v, sum v * f) over each 32-row tile
  const float* x2; long ldx2;   // f = x2[m, n] (row stride ldx2); NULL: f = v
  const float* gamma;           // with tile_totals: weights of the columns
  double* tile_totals;          // [ceil(M/32)][ceil(N/32)][2]: the two sums weighted by gamma, added over a tile's 32 columns
} vrnet_conv_colstats;
}

namespace {

struct IgemmArgs {
  const float* a; long lda;
  const float* w;
  const float* bias;
  float* y; long ldy;
  float* ypre; long ldypre;
  const float* res; long ldres; const float* res_scale;
  const float* kscale;
  const float* aux; long ldaux;
  int M, MH, MW, SH, SW, CK, CN;
  int kh, kw, stride, pad, dil;
  int mode, act, a_vec, b_vec, e_vec;
  int out_nchw, out_ctot, out_coff, accumulate;
  long wtap;   // Cout*Cin
  int Cin;
  int perm2;   // data gradient of a stride-2 conv: GEMM rows enumerate the 4 pixel parity classes one after another
  double* stats;   // [ceil(M/32)][ceil(CN/32)][2]: (sum, sum of squares) of the stored outputs per 32x32 tile, or NULL
  int stats_nb;    // ceil(CN/32)
  // Two-stream launch (the image and the radar chain of a backbone stage as ONE batch of 2B samples,
  // vr_coc.py:589-600): GEMM rows >= pair_rows belong to the second stream and use its own parameter set.
  // pair_rows is a multiple of every row-tile size, so a workgroup never straddles the two halves.
  int pair_rows;   // 0 = single parameter set
  const float* w2; const float* bias2; const float* res_scale2; const float* kscale2;
  // Column statistics of the stored outputs per 32-row tile (vector epilogue only): col_part[mb][n] = (sum_m v, sum_m v * f)
  // with f = col_x2[m, n] or, without col_x2, v itself -- the chunk partials of BatchNorm's batch statistics (forward) and
  // of GroupNorm's backward moments, which otherwise cost a pass over the tensor each; col_tot[mb][n / 32] = the same two
  // sums weighted by col_gamma[n] and added over the tile's 32 columns (GroupNorm backward: per-sample coefficients).
  double* col_part; const float* col_x2; long ld_col_x2; const float* col_gamma; double* col_tot;
  int dbg_fake_presplit;   // diagnostic build only (timing experiment, igemm.hip)
  // Output as bf16 planes (pgemm.hip; vector epilogue only): plane q of element (m, n) at yp[q * yp_plane + m * ldyp + n].
  // yp_np = 3: the stored fp32 value split exactly into three bf16 values (v = p0 + p1 + p2, round-to-nearest-even splits) --
  // the operand format of the next x6 GEMM, which then splits nothing; yp_np = 1: the value rounded to bf16.  `y` may be
  // null when only the planes are wanted.
  unsigned short* yp; long ldyp; long yp_plane; int yp_np;
  // Split contraction (round 4; the tile kernels of igemm.hip): workgroup (tile, split) runs steps [split, split + 1) * nsteps
  // / ksplit of the tile's K loop and stores its raw accumulators into slab `split` of kslab ([ksplit][grid tiles][4 waves]
  // [accumulators of the wave][64 lanes][16] floats); igemm_splitk_finish_kernel adds the slabs in order (deterministic) and
  // runs the epilogue.  For the small-M layers (16 x 16 maps: 2048 rows) whose tile grid leaves most CUs empty.
  int ksplit; float* kslab;
  // bf16 side tensors of the plane GEMMs on bf16 tensors (pgemm.hip, compute_dtype "bf16"): the pre-activation copy `ypre` is
  // STORED as bf16 and / or the GELU' argument `aux` is READ as bf16 (row strides in elements) -- the Mlp's u of the blocks whose
  // fc1 / fc2 are two launches, as the fused Mlp kernels keep it at precision 4
  int ypre_bf16, aux_bf16;
};

// The argument block a workgroup whose first row is m0 works with: the second stream's parameters behind pair_rows.
__device__ __forceinline__ IgemmArgs igemm_select_stream(const IgemmArgs& in, int m0) {
  IgemmArgs p = in;
  // parity-major rows (perm2): each of the 4 parity classes lists all samples in order, so the second stream is the
  // second half of every class (pair_rows is then half a class: M / 8)
  const int r0 = in.perm2 ? m0 % (in.M >> 2) : m0;
  if (in.pair_rows && r0 >= in.pair_rows) {
    p.w = in.w2; p.bias = in.bias2; p.res_scale = in.res_scale2; p.kscale = in.kscale2;
  }
  return p;
}

// GEMM row -> (sample, y, x) of the M-side pixel grid.  With perm2 the rows are parity-major: class (y&1, x&1)
// occupies a contiguous quarter of the rows, so every 64-row tile has ONE parity and the live-tap list drops
// the 5-8 of 9 taps that a stride-2 data gradient never touches for that class (instead of multiplying zeros).
__device__ __forceinline__ void igemm_row_to_pixel(const IgemmArgs& p, int m, int& b, int& y, int& x) {
  if (p.perm2) {
    const int Hh = p.MH >> 1, Wh = p.MW >> 1;
    const int per = (p.M >> 2);
    const int ph = m / per, r = m - ph * per;
    const int xh = r % Wh, q = r / Wh;
    const int yh = q % Hh;
    b = q / Hh;
    y = 2 * yh + (ph >> 1);
    x = 2 * xh + (ph & 1);
  } else {
    x = m % p.MW;
    const int q = m / p.MW;
    y = q % p.MH;
    b = q / p.MH;
  }
}
__device__ __forceinline__ long igemm_row_index(const IgemmArgs& p, int m) {
  if (!p.perm2) return m;
  int b, y, x;
  igemm_row_to_pixel(p, m, b, y, x);
  return ((long)b * p.MH + y) * p.MW + x;
}

constexpr int STAGE_LD = 36;                       // epilogue staging tile: 32 rows x 36 floats per wave

// One 32 x 32 accumulator tile whose first output element is (row0, col0); `stage`: this wave's private staging tile of
// 32 * STAGE_LD floats.
__device__ __forceinline__ void igemm_epilogue_tile(const IgemmArgs& p, const f32x16 acc, float* stage, int row0, int col0) {
  const int lane = threadIdx.x & 63;
  const int khalf = lane >> 5;
  // Row-contiguous float4 epilogue: each wave transposes one 32x32 accumulator tile through its private
  // LDS staging tile, then 8 lanes cover one 128-byte output row segment (aux / residual loads and all
  // stores are whole lines).  The main loop ended with a barrier, so the operand images can be reused.
#pragma unroll
  for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * khalf) * STAGE_LD + (lane & 31)] = acc[r];
  // the staging tile is private to this wave and a wave's LDS operations complete in program order: no workgroup
  // barrier, only a fence against compiler reordering
  __builtin_amdgcn_wave_barrier();
  // statistics of the stored values, accumulated per thread in fp64.  (An fp32 form -- deviations from the thread's first
  // value, un-shifted at the end -- was SLP-vectorised by hipcc into v_pk_*_f32 sequences with op_sel lane swaps whose
  // sum of squares came out short in ~1 launch of 30 whenever another kernel shared the CU: tools/debug/stats_race_dbg.py.
  // The same source built with -fno-slp-vectorize, or this fp64 form, is repeatable; the library is built without the
  // SLP vectoriser for that reason, see the Makefile.)
  double q1 = 0.0, q2 = 0.0;
  double cs1[4] = {0.0, 0.0, 0.0, 0.0}, cs2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int q = lane + 64 * t;
    const int row = q >> 3, c4 = (q & 7) * 4;
    const int m = row0 + row;
    const int n = col0 + c4;
    if (m < p.M && n < p.CN) {
      const long mo = igemm_row_index(p, m);
      f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * STAGE_LD + c4]);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.aux) {
        f32x4 a;
        if (p.aux_bf16) {
          const vr_bf16x4 h = *reinterpret_cast<const vr_bf16x4*>(reinterpret_cast<const unsigned short*>(p.aux) + mo * p.ldaux + n);
          a = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
          a = *reinterpret_cast<const f32x4*>(p.aux + mo * p.ldaux + n);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= vr_gelu_grad(a[e]);
      }
      if (p.ypre) {
        if (p.ypre_bf16) vr_store_planes4(reinterpret_cast<unsigned short*>(p.ypre) + mo * p.ldypre + n, 0, 1, v);
        else *reinterpret_cast<f32x4*>(p.ypre + mo * p.ldypre + n) = v;
      }
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = vr_gelu(v[e]);
      }
      if (p.res) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + mo * p.ldres + n);
        if (p.res_scale) v = rv + *reinterpret_cast<const f32x4*>(p.res_scale + n) * v;
        else v = rv + v;
      }
      if (p.y) {
        f32x4* dst = reinterpret_cast<f32x4*>(p.y + mo * p.ldy + n);
        if (p.accumulate) v += *dst;
        *dst = v;
      }
      if (p.yp) vr_store_planes4(p.yp + mo * p.ldyp + n, p.yp_plane, p.yp_np, v);
      if (p.stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double dv = (double)v[e];
          q1 += dv;
          q2 = __builtin_fma(dv, dv, q2);
        }
      }
      if (p.col_part) {
        f32x4 f = v;
        if (p.col_x2) f = *reinterpret_cast<const f32x4*>(p.col_x2 + mo * p.ld_col_x2 + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double dv = (double)v[e];
          cs1[e] += dv;
          cs2[e] = __builtin_fma(dv, (double)f[e], cs2[e]);
        }
      }
    }
  }
  if (p.col_part) {      // lane (row group lane >> 3, column quad lane & 7): add the 8 row groups, lanes 0-7 report
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        cs1[e] += __shfl_xor(cs1[e], o, 64);
        cs2[e] += __shfl_xor(cs2[e], o, 64);
      }
    }
    const int mb = row0 >> 5;
    const int n = col0 + (lane & 7) * 4;
    double g1 = 0.0, g2 = 0.0;
    if (lane < 8 && mb * 32 < p.M && n < p.CN) {
      double* d = p.col_part + ((long)mb * p.CN + n) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d[2 * e] = cs1[e];
        d[2 * e + 1] = cs2[e];
        if (p.col_tot) {
          const double g = (double)p.col_gamma[n + e];
          g1 += g * cs1[e];
          g2 += g * cs2[e];
        }
      }
    }
    if (p.col_tot) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        g1 += __shfl_xor(g1, o, 64);
        g2 += __shfl_xor(g2, o, 64);
      }
      const int nb = col0 >> 5;
      if (lane == 0 && mb * 32 < p.M && nb < p.stats_nb) {
        double* d = p.col_tot + ((long)mb * p.stats_nb + nb) * 2;
        d[0] = g1;
        d[1] = g2;
      }
    }
  }
  if (p.stats) {      // statistics of exactly what was stored (fp64, as the moments kernel): GroupNorm of the consumer
    const double st1 = wave_sum(q1), st2 = wave_sum(q2);
    const int mb = row0 >> 5, nb = col0 >> 5;
    if (lane == 0 && mb * 32 < p.M && nb < p.stats_nb) {
      double* d = p.stats + ((long)mb * p.stats_nb + nb) * 2;
      d[0] = st1;
      d[1] = st2;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// One 32x32 accumulator tile (tile row `ti`, tile column `tj` of the wave's TM x TN grid).  A function of ONE
// accumulator taken by value: looping `acc[i][j]` over runtime-looking indices (the compiler refuses to unroll a
// loop that contains barriers) makes the whole accumulator array runtime-indexed, i.e. moves it to scratch.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_vec(const IgemmArgs& p, const f32x16 acc, float* smem, int m0, int n0, int ti,
                                                   int tj) {
  const int wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  igemm_epilogue_tile(p, acc, smem + wave * (32 * STAGE_LD), m0 + wm * TM * 32 + ti * 32, n0 + wn * TN * 32 + tj * 32);
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_scalar(const IgemmArgs& p, const f32x16 acc, int m0, int n0, int ti, int tj) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int khalf = lane >> 5;
  const long hw = (long)p.MH * p.MW;
  const int n = n0 + wn * TN * 32 + tj * 32 + (lane & 31);
  if (n >= p.CN) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
  const float rsc = (p.res && p.res_scale) ? p.res_scale[n] : 1.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * TM * 32 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
    if (m >= p.M) continue;
    const long mo = igemm_row_index(p, m);
    float v = acc[r] + bias;
    if (p.aux) v *= vr_gelu_grad(p.aux[mo * p.ldaux + n]);
    if (p.ypre) p.ypre[mo * p.ldypre + n] = v;
    if (p.act == 1) v = fmaxf(v, 0.f);
    else if (p.act == 2) v = vr_gelu(v);
    if (p.res) v = p.res[mo * p.ldres + n] + rsc * v;
    float* dst;
    if (p.out_nchw) {
      const long b = mo / hw, pix = mo - b * hw;
      dst = p.y + ((b * p.out_ctot + p.out_coff + n) * hw + pix);
    } else {
      dst = p.y + mo * p.ldy + n;
    }
    if (p.accumulate) v += *dst;
    *dst = v;
  }
}

// Epilogue shared by the igemm kernels.  C/D map of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).  `smem` must hold 4 * 32 * STAGE_LD floats and no wave may
// still be reading operand images from it (callers end their main loop with a barrier).
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0) {
  static_assert((TM <= 2 && TN <= 2) || (TM == 1 && TN == 4), "accumulator tiles are named explicitly");
  if (p.e_vec) {
    igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[0][0], smem, m0, n0, 0, 0);
    if constexpr (TN > 1) igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[0][1], smem, m0, n0, 0, 1);
    if constexpr (TN > 2) {
      igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[0][2], smem, m0, n0, 0, 2);
      igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[0][3], smem, m0, n0, 0, 3);
    }
    if constexpr (TM > 1) {
      igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[1][0], smem, m0, n0, 1, 0);
      if constexpr (TN > 1) igemm_epilogue_vec<TM, TN, WM, WN>(p, acc[1][1], smem, m0, n0, 1, 1);
    }
    return;
  }
  igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[0][0], m0, n0, 0, 0);
  if constexpr (TN > 1) igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[0][1], m0, n0, 0, 1);
  if constexpr (TN > 2) {
    igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[0][2], m0, n0, 0, 2);
    igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[0][3], m0, n0, 0, 3);
  }
  if constexpr (TM > 1) {
    igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[1][0], m0, n0, 1, 0);
    if constexpr (TN > 1) igemm_epilogue_scalar<TM, TN, WM, WN>(p, acc[1][1], m0, n0, 1, 1);
  }
}

// ---- split contraction: raw accumulator slabs and the finishing kernel
template <int TM, int TN>
__device__ __forceinline__ void igemm_splitk_store(const IgemmArgs& p, const f32x16 (&acc)[TM][TN], long grid_tiles, int tile,
                                                   int split) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dst = p.kslab + (((long)split * grid_tiles + tile) * 4 + wave) * (TM * TN * 1024) + lane * 16;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *reinterpret_cast<f32x4*>(dst + (i * TN + j) * 1024 + 4 * q) = v;
      }
}

// One workgroup per tile (same tile map as the main kernel): sums the ksplit slabs in order and runs the fused epilogue.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_splitk_finish_kernel(const IgemmArgs p_in, int MT, int NT) {
  __shared__ __attribute__((aligned(16))) float smem[4 * 32 * STAGE_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = blockIdx.x, jj = L >> 3;
  const int nt = jj % NT, mt = (jj / NT) * 8 + (L & 7);
  if (mt >= MT) return;
  const int m0 = mt * (WM * TM * 32), n0 = nt * (WN * TN * 32);
  const IgemmArgs p = igemm_select_stream(p_in, m0);
  const long grid_tiles = gridDim.x;
  f32x16 acc[TM][TN];
  const float* src = p.kslab + ((long)L * 4 + wave) * (TM * TN * 1024) + lane * 16;
  const long sstride = grid_tiles * 4 * (TM * TN * 1024);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x4 t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const f32x4*>(src + (i * TN + j) * 1024 + 4 * q);
      for (int sp = 1; sp < p.ksplit; ++sp)
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] += *reinterpret_cast<const f32x4*>(src + sp * sstride + (i * TN + j) * 1024 + 4 * q);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = t[q][e];
    }
  igemm_epilogue<TM, TN, WM, WN>(p, acc, smem, m0, n0);
}

// Weight gradient: dW[t][n][c] = sum_m dy[m, n] * x[src(m, t), c], contraction over output pixels, split over row
// ranges into fp32 slabs (deterministic; reduced by wgrad_reduce_kernel).
struct WgradArgs {
  const float* x; long ldx;
  const float* dy; long lddy;
  float* slab; float* bslab;
  int M, OH, OW, H, W, Cin, Cout;
  int kh, kw, stride, pad, dil;
  int rows_per_split, n_tiles, c_tiles;
  int M_half;      // rows per stream: gridDim.z = 2 streams each contract their own M_half rows into their own slabs
  int splits;      // row splits per stream
  int xcd_group;   // x6 kernels: all tiles of a row split on one XCD (splits % 8 == 0), see wgrad_rows_xcd
};

// (first row, end row, slab index) of this workgroup: blockIdx.y = split within the stream, blockIdx.z = stream
__device__ __forceinline__ void wgrad_rows(const WgradArgs& p, int& m_begin, int& m_end, int& split) {
  const int z = blockIdx.z;
  const int base = z * p.M_half;
  m_begin = base + blockIdx.y * p.rows_per_split;
  const int limit = gridDim.z > 1 ? base + p.M_half : p.M;
  m_end = min(limit, m_begin + p.rows_per_split);
  split = z * gridDim.y + blockIdx.y;
}

// XCD-grouped variant (splits % 8 == 0): workgroups are dealt to the 8 XCDs round-robin in
// dispatch order, so dispatch ids j, j + 8, ... share an L2.  All tiles of one row split are mapped to ONE XCD: the
// split's dy / x rows then enter a single L2 once instead of once per XCD that holds one of its tiles.  `tile`
// replaces blockIdx.x.
__device__ __forceinline__ bool wgrad_rows_xcd(const WgradArgs& p, int& tile, int& m_begin, int& m_end, int& split) {
  const int tiles = gridDim.x;
  const int L = blockIdx.x + tiles * blockIdx.y, j = L >> 3;
  const int y = (j / tiles) * 8 + (L & 7);
  tile = j % tiles;
  if (y >= p.splits) return false;
  const int z = blockIdx.z;
  const int base = z * p.M_half;
  m_begin = base + y * p.rows_per_split;
  const int limit = gridDim.z > 1 ? base + p.M_half : p.M;
  m_end = min(limit, m_begin + p.rows_per_split);
  split = z * p.splits + y;
  return true;
}

}  // namespace
