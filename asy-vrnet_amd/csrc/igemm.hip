// Implicit-GEMM NHWC convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel family covers every dense convolution of the hot path (reference call sites:
// 1x1 fc1/fc_v/fc2 and Mlp fc1/fc2 -- backbone/fusion/vr_coc.py:145-147,205-207; 3x3 radar_projection
// :308; 3x3/s2 and 4x4/s4 PointRecuder :99-102; dilated ASPP -- neck/coc_fpn_dual.py:50-74; heads --
// head/decouplehead.py:21-40) in both directions:
//   mode 0 (forward)       y[m, n]  = sum_{t, c} x[src(m, t), c] * w[t][n][c]      m over output pixels
//   mode 1 (data gradient) dx[m, c] = sum_{t, n} dy[src'(m, t), n] * w[t][n][c]    m over input pixels
// Three kernels, chosen per shape by vrnet_conv2d_f32 (measured table: profiles/*per_shape_detail*):
//   igemm_kernel      register-staged (global -> registers -> k-major LDS images, ds_read_b32 fragments), 64 x 64 x 16
//                     tiles at 8 workgroups per CU (128 x {32,64,128} variants for narrow outputs / non-vector shapes);
//   igemm_dma_kernel  LDS-DMA ring (global_load_lds_dwordx4, 3 stages, XOR-swizzled ds_read_b128 fragments) for the
//                     layers whose grid cannot fill the chip at 8 workgroups per CU, and for long contractions;
//   igemm_bf16_kernel (igemm_bf16.hip) bf16-rounded operands on v_mfma_f32_32x32x16_bf16 (precision = 1).
// The epilogue (igemm_common.h) is shared.  Exact fp32: the MFMA is an fmaf chain in a fixed k order (guide:
// cdna_hip_programming.md section 3).  The weight gradient (wgrad_* below) contracts over pixels into fp32 slabs.
#include "igemm_common.h"
#include "x6.h"

#include <cstdlib>

// default variants of igemm_planes_reg_kernel for the 128 x 64 / 128 x 128 tile classes (0: igemm_planes_kernel; -1: per-shape rule)
#ifndef VR_PLANES_REG21
#define VR_PLANES_REG21 0
#endif
#ifndef VR_PLANES_STREAM_MIN_ROWS
#define VR_PLANES_STREAM_MIN_ROWS 0      // 0: never
#endif
#ifndef VR_PLANES_REG22
#define VR_PLANES_REG22 0
#endif

namespace {

template <int BM, int BN, int BK, int TM, int TN, int WM, int WN, int MODE, bool VEC>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64 && BK == 16) ? 8 : 1) void igemm_kernel(const IgemmArgs p_in) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile");
  const IgemmArgs p = igemm_select_stream(p_in, blockIdx.x * BM);
  constexpr int KQ = BK / 4;            // k-quads per row
  constexpr int RPP = 256 / KQ;         // rows covered per pass of the transposing loaders
  constexpr int AS_FLOATS = BK * (BM + 4), BS_FLOATS = BK * (BN + 4);
  constexpr int SM_FLOATS = (AS_FLOATS + BS_FLOATS) > 4 * 32 * STAGE_LD ? (AS_FLOATS + BS_FLOATS) : 4 * 32 * STAGE_LD;
  __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
  __shared__ unsigned tapmask_s;
  __shared__ unsigned char taps_s[32];
  float (*As)[BM + 4] = reinterpret_cast<float (*)[BM + 4]>(smem);
  float (*Bs)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(smem + AS_FLOATS);
  constexpr int AROWS = BM / RPP;                     // A rows per thread
  constexpr int BROWS = (BN >= RPP) ? BN / RPP : 1;   // NK loader (MODE 0): rows per thread
  constexpr int BVEC = (BN * BK / 4 + 255) / 256;     // KN loader (MODE 1): float4 per thread
  constexpr int BSLOTS = MODE == 0 ? BROWS : BVEC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int nkb = (p.CK + BK - 1) / BK;
  const int T = p.kh * p.kw;
  const int kq = tid % KQ, rbase = tid / KQ;

  // ---- A-side: each thread owns rows rbase [+RPP...], k-quad kq.  Pixel decomposition once per kernel.
  int a_b[AROWS], a_y[AROWS], a_x[AROWS];
  bool a_ok[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + rbase + RPP * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    igemm_row_to_pixel(p, mm, a_b[i], a_y[i], a_x[i]);
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
  // Taps that are invalid for every row of this tile (padding of dilated convs on small maps, the stride
  // phases of a strided data gradient) are dropped from the K loop: block-uniform list of live taps.
  int ntaps = T;
  const bool use_list = T > 1 && T <= 32;
  if (use_list) {
    if (tid == 0) tapmask_s = 0u;
    __syncthreads();
    unsigned mine = 0u;
    for (int t = 0; t < T; ++t) {
      const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
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

  // ---- per-tap operand pointers (hoisted out of the K loop): row bases, nullptr = masked
  const float* a_ptr[AROWS];
  const float* b_ptr[BSLOTS];
  auto setup_tap = [&](int t) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int sy, sx;
      const bool ok = src_of(i, ky, kx, sy, sx);
      a_ptr[i] = ok ? p.a + ((long)(a_b[i] * p.SH + sy) * p.SW + sx) * p.lda : nullptr;
    }
    const float* wt = p.w + (long)t * p.wtap;
    if (MODE == 0) {              // B[k = c][n] = w[t][n][c]: rows n, contiguous contraction
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        const int n = rbase + RPP * i;
        b_ptr[i] = (n < BN && n0 + n < p.CN) ? wt + (long)(n0 + n) * p.Cin : nullptr;
      }
    } else {                      // B[k = n'][j = c] = w[t][n'][c]: rows = contraction, contiguous output col
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        const int idx = tid + 256 * i;
        const int kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
        const int col = n0 + 4 * cq;
        b_ptr[i] = (kr < BK && col < p.CN) ? wt + (long)kr * p.Cin + col : nullptr;
      }
    }
  };

  f32x4 areg[AROWS], breg[BSLOTS];
  int ld_ti = 0, ld_kb = 0;        // (tap index, k block) of the NEXT tile to load
  auto load_tiles = [&]() {
    if (ld_kb == 0) setup_tap(use_list ? (int)taps_s[ld_ti] : ld_ti);
    const int c0 = ld_kb * BK;
    const int kc = c0 + 4 * kq;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ptr[i] != nullptr && kc < p.CK) {
        if (VEC) {
          v = *reinterpret_cast<const f32x4*>(a_ptr[i] + kc);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kc + j < p.CK) v[j] = a_ptr[i][kc + j];
        }
        if (MODE == 1 && p.kscale) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kc + j < p.CK) v[j] *= p.kscale[kc + j];
        }
      }
      areg[i] = v;
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b_ptr[i] != nullptr && kc < p.CK) {
          if (VEC) {
            v = *reinterpret_cast<const f32x4*>(b_ptr[i] + kc);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (kc + j < p.CK) v[j] = b_ptr[i][kc + j];
          }
        }
        breg[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        const int idx = tid + 256 * i;
        const int kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b_ptr[i] != nullptr && c0 + kr < p.CK) {
          const float* src = b_ptr[i] + (long)c0 * p.Cin;
          if (VEC) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
            const int col = n0 + 4 * cq;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < p.CN) v[j] = src[j];
          }
        }
        breg[i] = v;
      }
    }
    if (++ld_kb == nkb) {
      ld_kb = 0;
      ++ld_ti;
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = rbase + RPP * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) As[4 * kq + j][r] = areg[i][j];
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        const int n = rbase + RPP * i;
        if (n < BN) {
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[4 * kq + j][n] = breg[i][j];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i) {
        const int idx = tid + 256 * i;
        const int kr = idx / (BN / 4), cq = idx - kr * (BN / 4);
        if (kr < BK) *reinterpret_cast<f32x4*>(&Bs[kr][4 * cq]) = breg[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int arow = wm * TM * 32 + (lane & 31);
  const int bcol = wn * TN * 32 + (lane & 31);
  const int khalf = lane >> 5;

  if (nsteps > 0) load_tiles();
  for (int s = 0; s < nsteps; ++s) {
    store_tiles();
    __syncthreads();
    if (s + 1 < nsteps) load_tiles();
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[2 * kk + khalf][arow + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[2 * kk + khalf][bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  igemm_epilogue<TM, TN, WM, WN>(p, acc, smem, m0, n0);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the same implicit GEMM (64 x 64 x 32 tile, 4 waves, one 32x32 accumulator per wave) for
// the vector-aligned layers (all of the backbone / neck 1x1 and kxk convs): operand tiles go global -> LDS by
// `global_load_lds_dwordx4` into a ring of NST stage buffers with NST-1 stages in flight across the (raw) barrier
// of each K step, so a workgroup keeps 2 x 16 KB of loads outstanding without spending a register on them; the
// register-staged kernel above has one 8 KB tile in flight and needs ~8 workgroups per CU to cover HBM latency,
// which the small-M layers (M = 2048 / 8192 pixels: <= 3 waves per SIMD over the whole chip) never reach.
//   * K-contiguous operands (A always; B = w[n][k] in mode 0) are stored as rows of 8 quads (32 floats, one
//     128-B line), slot (r, q ^ ((r >> 1) & 7)): the XOR is applied to the per-lane SOURCE address (the DMA's LDS
//     destination is lane-linear) and again on the read, which makes every fragment ds_read_b128 conflict-free.
//     Lane (row, h) owns k = 16 h .. 16 h + 15 of the 32-deep stage; MFMA step i contracts k = i and k = 16 + i.
//   * mode 1's B = w[k][c] (contraction-major rows, contiguous output channels) is stored linearly as [32][64]
//     and read with conflict-free ds_read_b32; kscale (layer scale) is applied to those fragments from an LDS copy.
//   * out-of-range rows / taps / channel quads read a zero page instead of being masked (a masked DMA lane would
//     leave stale LDS behind).
//   * workgroup -> tile map: the NT column tiles of one row tile are consecutive on ONE XCD (ids b, b + 8, ...
//     share an XCD), so the A rows are fetched into a single L2 once.
__device__ __attribute__((aligned(128))) float vr_zero_page[64];

// T = 1: 64 x 64 x 32 tile, one 32x32 accumulator per wave (16 MFMAs per stage); T = 2: 128 x 128 x 16 tile, 2 x 2
// accumulators per wave (32 MFMAs per stage, half the LDS-fill bytes and fragment reads per MFMA) for the layers
// whose grid still fills the chip with 128-row tiles.  Both stage 16 KB per K step.
// Asymmetric tiles (TM != TN, e.g. 128 x 64 x 16) serve outputs of 64 / 192 / 320 channels and grids that 128 x 128
// tiles would leave half empty.
// (128 x 64 x 32 stages -- half the barriers / waits per MFMA -- were measured for launches with <= 2 workgroups per CU:
// +5..14 % on such launches alone, -0.5 % on the step, where the second chain's kernels already fill those CUs.)
template <int MODE, int NST, int TM, int TN, int PROD = 0>      // PROD: 0 fp32 MFMA, 6 x6, 1 bf16-rounded operands
__global__ __launch_bounds__(256, NST <= 3 ? 3 : (NST <= 4 ? 2 : 1)) void igemm_dma_kernel(const IgemmArgs p_in, int MT, int NT) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = (TM == 1 && TN == 1) ? 32 : 16;
  constexpr int QPR = BK / 4;                       // 16-byte quads per K-contiguous row of a stage
  constexpr int KQ = QPR / 2;                       // quads each lane owns per stage (k = h*BK/2 .. +BK/2)
  constexpr int RSH = QPR == 8 ? 1 : 2;             // swizzle: quad' = quad ^ ((row >> RSH) & (QPR - 1))
  constexpr int A_FLOATS = BM * BK, ST_FLOATS = A_FLOATS + BN * BK;
  constexpr int NA = A_FLOATS / 1024, NB = BN * BK / 1024, NP = NA + NB;      // 1 KB DMA pieces per wave and stage
  constexpr int RING = NST * ST_FLOATS;
  constexpr int KS_MAX = 1024;                      // kscale copy (mode 1)
  static_assert(RING >= 4 * 32 * STAGE_LD, "epilogue staging must fit the ring");
  static_assert(NST >= 3 && NST <= 6, "counted waits cover up to 4 younger stages");
  static_assert(NA >= 1 && NB >= 1 && NA * 1024 == A_FLOATS && NB * 1024 == BN * BK, "whole 1 KB DMA pieces per wave");
  static_assert(NP == 4 || NST == 3, "the counted waits of deeper rings assume 4 pieces per stage");
  // one LDS object only: a second __shared__ beside a DMA staging array makes hipcc drain vmcnt before ds_reads
  __shared__ __attribute__((aligned(16))) float smem[RING + 16 + (MODE == 1 ? KS_MAX : 0)];
  unsigned* tapmask_s = reinterpret_cast<unsigned*>(smem + RING);
  unsigned char* taps_s = reinterpret_cast<unsigned char*>(smem + RING + 4);     // 32 bytes
  float* ks_s = smem + RING + 16;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave id as a scalar: the LDS destinations of the DMAs (M0) then come from SALU arithmetic, not from a
  // v_readfirstlane per piece
  // The zero page's address is taken ONCE and made opaque: left to itself hipcc re-derives it from the GOT
  // (s_getpc + s_load_dwordx2 + s_waitcnt lgkmcnt(0)) in front of EVERY DMA of the main loop, and that wait also
  // drains the fragment ds_reads in flight.
  const float* zero_page = vr_zero_page;
  asm volatile("" : "+s"(zero_page));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int GT = 8 * ((MT + 7) >> 3) * NT;            // workgroups per split (the tile map pads row tiles to eights)
  const int split = __builtin_amdgcn_readfirstlane(p_in.ksplit > 1 ? (int)blockIdx.x / GT : 0);
  const int L = blockIdx.x - split * GT, jj = L >> 3;
  const int nt = jj % NT, mt = (jj / NT) * 8 + (L & 7);
  if (mt >= MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const IgemmArgs p = igemm_select_stream(p_in, m0);
#ifdef VR_IGEMM_STAMP2
  // diagnostic build: per-workgroup timeline [start, first stage landed, main loop done, epilogue done, HW id]
  unsigned long long* wg_stamp = reinterpret_cast<unsigned long long*>(p_in.stats) + 8 * (long)blockIdx.x;
  if (tid == 0) {
    wg_stamp[0] = __builtin_amdgcn_s_memtime();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    wg_stamp[4] = hw;
    wg_stamp[5] = __builtin_readcyclecounter();
  }
#endif
  const int nkb = (p.CK + BK - 1) / BK;
  const int TAPS = p.kh * p.kw;

  // ---- loader roles: two 16-B slots of the A image and two of the B image per thread and stage
  int a_q[NA], a_b[NA], a_y[NA], a_x[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int sl = (wave * NA + i) * 64 + lane, r = sl / QPR;
    a_q[i] = (sl % QPR) ^ ((r >> RSH) & (QPR - 1));
    const int m = m0 + r;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    igemm_row_to_pixel(p, mm, a_b[i], a_y[i], a_x[i]);
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
  int ntaps = TAPS;
  const bool use_list = TAPS > 1 && TAPS <= 32;
  if (use_list) {      // block-uniform list of the taps that are live for at least one row of this tile
    if (tid == 0) *tapmask_s = 0u;
    __syncthreads();
    unsigned mine = 0u;
    for (int t = 0; t < TAPS; ++t) {
      const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int sy, sx;
        if (src_of(i, ky, kx, sy, sx)) mine |= 1u << t;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o, 64);
    if (lane == 0 && mine) atomicOr(tapmask_s, mine);
    __syncthreads();
    if (tid == 0) {
      const unsigned mk = *tapmask_s;
      int c = 0;
      for (int t = 0; t < TAPS; ++t)
        if (mk & (1u << t)) taps_s[c++] = (unsigned char)t;
    }
    __syncthreads();
    ntaps = __popc(*tapmask_s);
  }
  if (MODE == 1 && p.kscale) {
    for (int i = tid; i < nkb * BK; i += 256) ks_s[i] = i < p.CK ? p.kscale[i] : 0.f;   // padded: the loop reads it unguarded
    __syncthreads();     // ordinary loads retire here, before the first DMA is issued
  }
  // split contraction: this workgroup runs steps [s_begin, s_begin + nsteps) of the tile's ntaps * nkb
  const int nsteps_all = ntaps * nkb;
  const int s_begin = p.ksplit > 1 ? (int)((long)split * nsteps_all / p.ksplit) : 0;
  const int nsteps = (p.ksplit > 1 ? (int)((long)(split + 1) * nsteps_all / p.ksplit) : nsteps_all) - s_begin;

  // Per tap: a running source pointer and a per-stage advance for each of the thread's four DMA slots.  Slots that
  // are masked for the whole tap (row outside the tile / image, padded tap, channel quad beyond CK or CN) point at
  // the zero page and do not advance, so a stage issues with two 64-bit adds per slot and no compares; only the
  // last K block of a tap whose contraction is not a multiple of BK re-checks the quad against CK.
  const float* a_run[NA];
  const float* b_run[NB];
  int a_inc[NA], b_inc[NB], a_k[NA], b_k[NB];
  const bool k_tail = (p.CK % BK) != 0;
  auto setup_tap = [&](int t) {
    const int ky = t / p.kw, kx = t - ky * p.kw;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int sy, sx;
      const bool ok = src_of(i, ky, kx, sy, sx) && 4 * a_q[i] < p.CK;
      a_run[i] = ok ? p.a + ((long)(a_b[i] * p.SH + sy) * p.SW + sx) * p.lda + 4 * a_q[i] : zero_page;
      a_inc[i] = ok ? BK : 0;
      a_k[i] = 4 * a_q[i];
    }
    const float* wt = p.w + (long)t * p.wtap;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int sl = (wave * NB + i) * 64 + lane;
      if (MODE == 0) {           // rows n, contiguous contraction: same image as A
        const int r = sl / QPR, q = (sl % QPR) ^ ((r >> RSH) & (QPR - 1));
        const bool ok = n0 + r < p.CN && 4 * q < p.CK;
        b_run[i] = ok ? wt + (long)(n0 + r) * p.Cin + 4 * q : zero_page;
        b_inc[i] = ok ? BK : 0;
        b_k[i] = 4 * q;
      } else {                   // rows = contraction index, contiguous output channels: linear [BK][BN]
        const int kr = sl / (BN / 4), col = n0 + 4 * (sl % (BN / 4));
        const bool ok = col < p.CN && kr < p.CK;
        b_run[i] = ok ? wt + (long)kr * p.Cin + col : zero_page;
        b_inc[i] = ok ? BK * p.Cin : 0;
        b_k[i] = kr;
      }
    }
  };

  int ld_ti = s_begin / nkb, ld_kb = s_begin - (s_begin / nkb) * nkb, ld_buf = 0;       // (tap index, k block, ring slot) of the NEXT stage to issue
  if (ld_kb != 0 && nsteps > 0) {      // a split that starts inside a tap: set the tap up and advance to its k block
    setup_tap(use_list ? (int)taps_s[ld_ti] : ld_ti);
#pragma unroll
    for (int i = 0; i < NA; ++i) a_run[i] += (long)ld_kb * a_inc[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) b_run[i] += (long)ld_kb * b_inc[i];
  }
  // The four DMA instructions of a stage are issued one by one (piece 0, 1 = A, 2, 3 = B): in the main loop each one
  // goes behind a group of MFMAs, whose 64-cycle execution hides the DMA's issue cost (60-185 cycles per piece when
  // issued back to back in front of the fragment reads).
  auto issue_begin = [&]() {
    if (ld_kb == 0) setup_tap(use_list ? (int)taps_s[ld_ti] : ld_ti);
  };
  auto issue_piece = [&](int i) {
    float* stage = smem + ld_buf * ST_FLOATS;
    const bool last = k_tail && ld_kb == nkb - 1;          // block-uniform
    const int c0 = ld_kb * BK;
    if (i < NA) {
      const float* src = (last && c0 + a_k[i] >= p.CK) ? zero_page : a_run[i];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + (wave * NA + i) * 256), 16, 0, 0);
      a_run[i] += a_inc[i];
    } else {
      const int j = i - NA;
      const float* src = (last && c0 + b_k[j] >= p.CK) ? zero_page : b_run[j];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + A_FLOATS + (wave * NB + j) * 256),
                                       16, 0, 0);
      b_run[j] += b_inc[j];
    }
  };
  auto issue_end = [&]() {
    if (++ld_kb == nkb) {
      ld_kb = 0;
      ++ld_ti;
    }
    if (++ld_buf == NST) ld_buf = 0;
  };
  auto issue = [&]() {
    issue_begin();
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_piece(i);
    issue_end();
  };

  // (one accumulator chain per tile: a second, independent set was measured with in-kernel stamps and changes
  // nothing -- with one wave per SIMD the 16 MFMAs of a stage take 1540 cycles because the wave's four DMA issues
  // cost ~130 cycles each in its own instruction stream, not because of the accumulator dependency)
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int h = lane >> 5;
  int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ra = wm * 32 * TM + 32 * i + (lane & 31);
    a_off[i] = ra * BK; a_swz[i] = (ra >> RSH) & (QPR - 1);
  }
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int rb = wn * 32 * TN + 32 * i + (lane & 31);
    b_off[i] = MODE == 0 ? rb * BK : rb; b_swz[i] = (rb >> RSH) & (QPR - 1);
  }

#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nsteps) issue();
  int cur = 0, kb = s_begin - (s_begin / nkb) * nkb;
#ifdef VR_IGEMM_STAMP
  unsigned long long* stamp = reinterpret_cast<unsigned long long*>(p.stats);     // diagnostic build: 64 x 4 stamps
#endif
  for (int s = 0; s < nsteps; ++s) {
#ifdef VR_IGEMM_STAMP
    if (blockIdx.x == 8 && tid == 0 && s < 64) stamp[4 * s + 3] = __builtin_amdgcn_s_memtime();
#endif
    // stage s has landed once at most the min(NST - 2, stages left) younger stages (4 DMAs each) are still outstanding
    {
      const int younger = nsteps - 1 - s < NST - 2 ? nsteps - 1 - s : NST - 2;      // block-uniform
      if (NP == 4 && younger >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (NP == 4 && younger == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (NP == 4 && younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NP == 4 && younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NP == 3 && younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (NP == 5 && younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (NP == 6 && younger == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef VR_IGEMM_STAMP2
    if (tid == 0 && s == 0) wg_stamp[1] = __builtin_amdgcn_s_memtime();
#endif
#ifdef VR_IGEMM_STAMP
    if (blockIdx.x == 8 && tid == 0 && s < 64) stamp[4 * s + 0] = __builtin_amdgcn_s_memtime();
#endif
    const bool more = s + NST - 1 < nsteps;   // next stage goes into the slot every wave finished reading before this barrier
    if (more) issue_begin();
    const float* As = smem + cur * ST_FLOATS;
    const float* Bs = As + A_FLOATS;
    f32x4 af[TM][KQ], bq[TN][KQ];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < KQ; ++j) af[i][j] = *reinterpret_cast<const f32x4*>(As + a_off[i] + 4 * ((KQ * h + j) ^ a_swz[i]));
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < KQ; ++j) bq[i][j] = *reinterpret_cast<const f32x4*>(Bs + b_off[i] + 4 * ((KQ * h + j) ^ b_swz[i]));
    } else {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < KQ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[i][j][e] = Bs[(4 * KQ * h + 4 * j + e) * BN + b_off[i]];
      if (p.kscale) {      // all quads of the (zero-padded) stash first: one LDS wait for the stage, not one per quad
        f32x4 ks[KQ];
#pragma unroll
        for (int j = 0; j < KQ; ++j) ks[j] = *reinterpret_cast<const f32x4*>(ks_s + kb * BK + 4 * KQ * h + 4 * j);
#pragma unroll
        for (int j = 0; j < KQ; ++j)
#pragma unroll
          for (int i = 0; i < TN; ++i) bq[i][j] *= ks[j];
      }
    }
#ifdef VR_IGEMM_STAMP
    if (blockIdx.x == 8 && tid == 0 && s < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp[4 * s + 1] = __builtin_amdgcn_s_memtime(); }
#endif
    constexpr int GROUPS = KQ >= 4 ? 4 : KQ;          // MFMA groups that each carry DMA pieces behind them
    if constexpr (PROD == 1) {
      constexpr int NK16 = KQ / 2;
      static_assert(KQ % 2 == 0, "bf16 MFMA needs whole k16 steps");
#pragma unroll
      for (int ks = 0; ks < NK16; ++ks) {
        vr_bf16x8 a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = vr_round8(af[i][2 * ks], af[i][2 * ks + 1]);
#pragma unroll
        for (int i = 0; i < TN; ++i) b1[i] = vr_round8(bq[i][2 * ks], bq[i][2 * ks + 1]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[jn], acc[i][jn], 0, 0, 0);
        if (more) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NP; ++q)
            if (q % NK16 == ks) issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if constexpr (PROD == 6) {
      // lane h owns k = (BK/2) h .. + BK/2 - 1 of the stage: 8 consecutive values per k16 step of the bf16 MFMA
      constexpr int NK16 = KQ / 2;
      static_assert(KQ % 2 == 0, "x6 needs whole k16 steps");
#pragma unroll
      for (int ks = 0; ks < NK16; ++ks) {
        vr_bf16x8 a3[TM][3], b3[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i) vr_split3(af[i][2 * ks], af[i][2 * ks + 1], a3[i]);
#ifdef VR_TUNING
        // timing experiment (results garbage): what would weights that arrive pre-split into bf16 planes be worth?
        // VRNET_X6_FAKE_PRESPLIT=1 takes the B fragments' bits as they are instead of splitting them.
        if (p.dbg_fake_presplit) {
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            b3[i][0] = __builtin_bit_cast(vr_bf16x8, bq[i][2 * ks]);
            b3[i][1] = __builtin_bit_cast(vr_bf16x8, bq[i][2 * ks + 1]);
            b3[i][2] = b3[i][0];
          }
        } else
#endif
#pragma unroll
        for (int i = 0; i < TN; ++i) vr_split3(bq[i][2 * ks], bq[i][2 * ks + 1], b3[i]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) acc[i][jn] = vr_mfma_x6(a3[i], b3[jn], acc[i][jn]);
        if (more) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NP; ++q)
            if (q % NK16 == ks) issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][j][e], bq[jn][j][e], acc[i][jn], 0, 0, 0);
      if (more && j < GROUPS) {
        __builtin_amdgcn_sched_barrier(0);          // keep the DMA behind this MFMA group, not hoisted to the top
#pragma unroll
        for (int q = 0; q < NP; ++q)
          if (q % GROUPS == j) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) issue_end();
#ifdef VR_IGEMM_STAMP
    if (blockIdx.x == 8 && tid == 0 && s < 64) stamp[4 * s + 2] = __builtin_amdgcn_s_memtime();
#endif
    if (++cur == NST) cur = 0;
    if (++kb == nkb) kb = 0;
  }
  __syncthreads();
#ifdef VR_IGEMM_STAMP2
  if (tid == 0) wg_stamp[2] = __builtin_amdgcn_s_memtime();
  {
    IgemmArgs q = p;
    q.stats = nullptr;
    igemm_epilogue<TM, TN, 2, 2>(q, acc, smem, m0, n0);
  }
  __syncthreads();
  if (tid == 0) wg_stamp[3] = __builtin_amdgcn_s_memtime();
  return;
#endif
#ifdef VR_IGEMM_STAMP
  IgemmArgs q = p;            // the stats pointer carries the stamps in this build
  q.stats = nullptr;
  igemm_epilogue<TM, TN, 2, 2>(q, acc, smem, m0, n0);
#else
  if (p.ksplit > 1) igemm_splitk_store<TM, TN>(p, acc, GT, L, split);
  else igemm_epilogue<TM, TN, 2, 2>(p, acc, smem, m0, n0);
#endif
}


// ------------------------------------------------------------------------------------------------
// x6 implicit GEMM for 1x1 convs with PRE-SPLIT weights (round 3).  The six-product scheme spends its VALU time on splitting
// fragments into bf16 planes, and the matrix pipe and the VALU do not overlap on gfx950 -- but the B operand is the same
// weight tile for every row tile of every launch of a step.  vrnet_conv_planes_pack_f32 splits every eligible weight ONCE
// per step into three bf16 planes laid out as the LDS image of a stage: [k16 step][64-column block][plane][lane half][64
// columns] x 8 bf16, so a stage's B tile is one contiguous 6 KB (x TN) block that goes global -> LDS by DMA as it is and
// every B fragment is one conflict-free ds_read_b128 per plane; only the A (activation) fragments are split in the
// kernel: 2 instead of 3 (128 x 64 tile) resp. 4 (128 x 128) fragment splits per K16 step.  Both directions use this
// kernel: the forward pack holds w[n][c] (columns n, contraction c), the data-gradient pack the transposed weights with
// the layer scale folded in (columns c, contraction n).  A operand, ring, XCD-aware tile map and epilogue as in
// igemm_dma_kernel.  Measured bound of the idea (diagnostic build, B splits skipped): 27.8 -> 26.6 ms per step.
// (Round 3 also had a variant that folded the GroupNorm in front of the conv into the A fragments -- bit-identical to launch +
// conv, measured neutral-to-negative in the step at every threshold, never on by default; removed in round 4.)
template <int TN, int NST>
__global__ __launch_bounds__(256, (NST * (8192 + TN * 6144) <= 49152) ? 3 : 2) void igemm_planes_kernel(const IgemmArgs p,
                                                                                                      const unsigned char* planes,
                                                                                                      int JB, int MT, int NT) {
  // Wave layout 4 x 1: a wave owns 32 of the tile's 128 rows and ALL its columns, so every A fragment is split by exactly
  // one wave (with the 2 x 2 layout of igemm_dma_kernel the two waves of a row pair split the same rows) -- with B free, the
  // A splits are the whole VALU cost: 1 split per 12 (128 x 64 tile) resp. 24 (128 x 128) MFMAs.
  constexpr int TM = 1, TNW = 2 * TN, BM = 128, BN = 64 * TN, BK = 16;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = TN * 6144, ST_BYTES = A_BYTES + B_BYTES;
  constexpr int NA = A_BYTES / 1024 / 4;                  // A pieces per wave and stage (2)
  constexpr int NBT = B_BYTES / 1024;                     // B pieces per stage: 6 (waves 0-1 issue 2, waves 2-3 one) or 12 (3 each)
  static_assert(NST * ST_BYTES >= 4 * 32 * STAGE_LD * 4, "epilogue staging must fit the ring");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * ST_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const float* zero_page = vr_zero_page;
  asm volatile("" : "+s"(zero_page));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int GT = 8 * ((MT + 7) >> 3) * NT;
  const int split = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? (int)blockIdx.x / GT : 0);
  const int L = blockIdx.x - split * GT, jj = L >> 3;
  const int nt = jj % NT, mt = (jj / NT) * 8 + (L & 7);
  if (mt >= MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nsteps_all = p.CK / BK;
  const int s_begin = p.ksplit > 1 ? (int)((long)split * nsteps_all / p.ksplit) : 0;
  const int nsteps = (p.ksplit > 1 ? (int)((long)(split + 1) * nsteps_all / p.ksplit) : nsteps_all) - s_begin;
  const int NBW = TN == 2 ? 3 : (wave < 2 ? 2 : 1);      // this wave's B pieces per stage (wave-uniform)

  // ---- A loader roles (as igemm_dma_kernel with QPR = 4): slot (row, quad) of a 128 x 16 fp32 image, XOR-swizzled
  const float* a_run[NA];
  int a_inc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int sl = (wave * NA + i) * 64 + lane, r = sl >> 2;
    const int q = (sl & 3) ^ ((r >> 2) & 3);
    const int m = m0 + r;
    const bool ok = m < p.M;
    a_run[i] = ok ? p.a + (long)m * p.lda + 4 * q + (long)s_begin * BK : zero_page;
    a_inc[i] = ok ? BK : 0;
  }
  const long b_step = (long)JB * 6144;
  const unsigned char* b_src = planes + ((long)(n0 >> 6)) * 6144 + (long)lane * 16 + s_begin * b_step;       // + kb * JB * 6144
  int ld_buf = 0;
  // one DMA piece of the stage being issued: 0 .. NA - 1 = A, then this wave's B pieces
  auto issue_piece = [&](int i) {
    unsigned char* stage = smem + ld_buf * ST_BYTES;
    if (i < NA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_run[i],
                                       (__attribute__((address_space(3))) void*)(stage + (wave * NA + i) * 1024), 16, 0, 0);
      a_run[i] += a_inc[i];
    } else if (i - NA < NBW) {
      const int j = i - NA;
      const int piece = TN == 2 ? wave * 3 + j : (wave < 2 ? wave * 2 + j : 2 + wave);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + piece * 1024),
                                       (__attribute__((address_space(3))) void*)(stage + A_BYTES + piece * 1024), 16, 0, 0);
    }
  };
  auto issue_end = [&]() {
    b_src += b_step;
    if (++ld_buf == NST) ld_buf = 0;
  };
  constexpr int NPMAX = NA + (TN == 2 ? 3 : 2);
  auto issue = [&]() {
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) issue_piece(i);
    issue_end();
  };

  f32x16 acc[TM][TNW];
#pragma unroll
  for (int j = 0; j < TNW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
  const int h = lane >> 5;
  const int ra = wave * 32 + (lane & 31);
  const int a_off = ra * BK * 4, a_swz = (ra >> 2) & 3;
  int b_off[TNW];
#pragma unroll
  for (int i = 0; i < TNW; ++i) {
    const int rb = 32 * i + (lane & 31);
    b_off[i] = ((rb >> 6) * 6 + h) * 1024 + (rb & 63) * 16;       // + plane * 2048
  }

#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nsteps) issue();
  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (NST >= 3 && nsteps - 1 - s >= 1) {       // one younger stage stays in flight
      if (TN == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (wave < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = s + NST - 1 < nsteps;      // the next stage goes into the slot every wave finished reading before this barrier
    const unsigned char* As = smem + cur * ST_BYTES;
    const unsigned char* Bs = As + A_BYTES;
    vr_bf16x8 a3[3], b3[TNW][3];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b3[i][pl] = *reinterpret_cast<const vr_bf16x8*>(Bs + b_off[i] + pl * 2048);
    {
      f32x4 lo = *reinterpret_cast<const f32x4*>(As + a_off + 16 * ((2 * h) ^ a_swz));
      f32x4 hi = *reinterpret_cast<const f32x4*>(As + a_off + 16 * ((2 * h + 1) ^ a_swz));
      vr_split3(lo, hi, a3);
    }
    // each DMA instruction of the next stage goes behind a group of six MFMAs, whose execution hides its issue cost
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn) {
      acc[0][jn] = vr_mfma_x6(a3, b3[jn], acc[0][jn]);
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NPMAX; ++q)
          if (q % TNW == jn) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) issue_end();
    if (++cur == NST) cur = 0;
  }
  __syncthreads();
  if (p.ksplit > 1) igemm_splitk_store<TM, TNW>(p, acc, GT, L, split);
  else igemm_epilogue<TM, TNW, 4, 1>(p, acc, reinterpret_cast<float*>(smem), m0, n0);
}

// Multi-tensor weight split: one launch for every eligible weight of a step.  Table entry e (8 longs): source address,
// columns J, contraction K, element strides (column, contraction), contraction-scale address or 0, destination address,
// first block.  Block = (entry, k16 step, 64-column block); thread = (lane half, column): 8 values -> 3 x 8 bf16.
__global__ __launch_bounds__(128) void planes_pack_kernel(const long* table, int nentries) {
  int e = 0;
  {   // the entry this block belongs to: first-block offsets ascend
    int lo = 0, hi = nentries - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[8 * mid + 7] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    e = lo;
  }
  const long* t = table + 8 * e;
  const float* src = reinterpret_cast<const float*>(t[0]);
  const int J = (int)t[1], K = (int)t[2];
  const long sj = t[3], sk = t[4];
  const float* ksc = reinterpret_cast<const float*>(t[5]);
  unsigned short* dst = reinterpret_cast<unsigned short*>(t[6]);
  const int JB = ((J + 127) >> 7) << 1;       // 64-column blocks, padded to whole 128-column tiles
  const int blk = blockIdx.x - (int)t[7];
  const int kb = blk / JB, jb = blk - kb * JB;
  const int hh = threadIdx.x >> 6, col = threadIdx.x & 63;
  const int j = jb * 64 + col;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = kb * 16 + 8 * hh + i;
    float x = (j < J && k < K) ? src[(long)j * sj + (long)k * sk] : 0.f;
    if (ksc && k < K) x *= ksc[k];
    v[i] = x;
  }
  unsigned short* d = dst + ((long)(kb * JB + jb) * 6144 + hh * 1024 + col * 16) / 2;      // plane stride 2048 B
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 p0 = (__bf16)v[i];
    const float r1 = v[i] - (float)p0;
    const __bf16 p1 = (__bf16)r1;
    const float r2 = r1 - (float)p1;
    const __bf16 p2 = (__bf16)r2;
    d[i] = __builtin_bit_cast(unsigned short, p0);
    d[1024 + i] = __builtin_bit_cast(unsigned short, p1);
    d[2048 + i] = __builtin_bit_cast(unsigned short, p2);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[t][n][c] = sum_m dy[m, n] * x[src(m, t), c], contraction over output pixels,
// split over `S` row ranges into fp32 slabs (deterministic; reduced by wgrad_reduce_kernel).

// IDENT: 1x1 / stride 1 / pad 0 -- the gathered x row of output pixel m is row m itself (no index math).
constexpr int BK = 16;   // wgrad contraction step

template <int BM, int BN, int TM, int TN, int WM, int WN, bool IDENT, bool VEC>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64) ? 8 : 1) void wgrad_kernel(const WgradArgs p) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile");
  __shared__ __attribute__((aligned(16))) float As[BK][BM + 4];   // dy tile, [m][n]
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + 4];   // gathered x tile, [m][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  const int ct = bid % p.c_tiles; bid /= p.c_tiles;
  const int nt = bid % p.n_tiles; bid /= p.n_tiles;
  const int t = bid;
  const int ky = t / p.kw, kx = t - ky * p.kw;
  const int n0 = nt * BM, c0 = ct * BN;
  int m_begin, m_end, split;
  wgrad_rows(p, m_begin, m_end, split);
  constexpr int AVEC = (BM * BK / 4 + 255) / 256;
  constexpr int BVEC = (BN * BK / 4 + 255) / 256;
  f32x4 areg[AVEC], breg[BVEC];
  const bool do_bias = p.bslab != nullptr && ct == 0 && t == 0;
  float bsum = 0.f;
  // per-thread fixed (row-in-tile, column) slots
  int a_kr[AVEC], a_col[AVEC], b_kr[BVEC], b_col[BVEC];
#pragma unroll
  for (int i = 0; i < AVEC; ++i) {
    const int idx = tid + 256 * i;
    a_kr[i] = idx / (BM / 4);
    a_col[i] = n0 + 4 * (idx - a_kr[i] * (BM / 4));
  }
#pragma unroll
  for (int i = 0; i < BVEC; ++i) {
    const int idx = tid + 256 * i;
    b_kr[i] = idx / (BN / 4);
    b_col[i] = c0 + 4 * (idx - b_kr[i] * (BN / 4));
  }

  auto load_tiles = [&](int mb) {
#pragma unroll
    for (int i = 0; i < AVEC; ++i) {
      const int m = mb + a_kr[i];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_kr[i] < BK && m < m_end && a_col[i] < p.Cout) {
        const float* src = p.dy + (long)m * p.lddy + a_col[i];
        if (VEC) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (a_col[i] + j < p.Cout) v[j] = src[j];
        }
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BVEC; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int m = mb + b_kr[i];
      if (b_kr[i] < BK && m < m_end && b_col[i] < p.Cin) {
        long row;
        bool ok = true;
        if (IDENT) {
          row = m;
        } else {
          const int ox = m % p.OW;
          const int q = m / p.OW;
          const int oy = q % p.OH, b = q / p.OH;
          const int sy = oy * p.stride - p.pad + ky * p.dil, sx = ox * p.stride - p.pad + kx * p.dil;
          ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
          row = (long)(b * p.H + sy) * p.W + sx;
        }
        if (ok) {
          const float* src = p.x + row * p.ldx + b_col[i];
          if (VEC) {
            v = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (b_col[i] + j < p.Cin) v[j] = src[j];
          }
        }
      }
      breg[i] = v;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AVEC; ++i)
      if (a_kr[i] < BK) *reinterpret_cast<f32x4*>(&As[a_kr[i]][a_col[i] - n0]) = areg[i];
#pragma unroll
    for (int i = 0; i < BVEC; ++i)
      if (b_kr[i] < BK) *reinterpret_cast<f32x4*>(&Bs[b_kr[i]][b_col[i] - c0]) = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int arow = wm * TM * 32 + (lane & 31);
  const int bcol = wn * TN * 32 + (lane & 31);
  const int khalf = lane >> 5;

  if (m_begin < m_end) load_tiles(m_begin);
  for (int mb = m_begin; mb < m_end; mb += BK) {
    store_tiles();
    __syncthreads();
    if (mb + BK < m_end) load_tiles(mb + BK);
    if (do_bias && tid < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) bsum += As[k][tid];
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[2 * kk + khalf][arow + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[2 * kk + khalf][bcol + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const long T = (long)p.kh * p.kw;
  float* slab = p.slab + ((long)split * T + t) * p.Cout * p.Cin;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = c0 + wn * TN * 32 + j * 32 + (lane & 31);
    if (c >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (n < p.Cout) slab[(long)n * p.Cin + c] = acc[i][j][r];
      }
  }
  if (do_bias && tid < BM && n0 + tid < p.Cout) p.bslab[(long)split * p.Cout + n0 + tid] = bsum;
}

// LDS-DMA variant of the weight gradient (64 x 64 tile of dW, 32 contraction rows per stage, ring of 3 stages with
// two in flight): both operand tiles are [32 rows m][64 channels] images, lane-linear, read by conflict-free
// ds_read_b32 (lanes = consecutive channels).  The split-over-rows grids of the small weight matrices put only
// 1-3 workgroups on a CU, which the one-tile-in-flight kernel above cannot turn into HBM bandwidth.
template <bool IDENT>
__global__ __launch_bounds__(256) void wgrad_dma_kernel(const WgradArgs p) {
  constexpr int BKD = 32, NST = 3, T_FLOATS = BKD * 64, ST_FLOATS = 2 * T_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[NST * ST_FLOATS];
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
  float bsum = 0.f;
  int s_kr[2], s_cq[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int sl = (wave * 2 + i) * 64 + lane;
    s_kr[i] = sl >> 4;
    s_cq[i] = 4 * (sl & 15);
  }
  const float* zero_page = vr_zero_page;          // taken once and opaque: see igemm_dma_kernel
  asm volatile("" : "+s"(zero_page));
  int ld_m = m_begin, ld_buf = 0;
  auto issue = [&]() {
    float* stage = smem + ld_buf * ST_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = ld_m + s_kr[i];
      const float* src = (m < m_end && n0 + s_cq[i] < p.Cout) ? p.dy + (long)m * p.lddy + n0 + s_cq[i] : zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + (wave * 2 + i) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = ld_m + s_kr[i];
      const float* src = zero_page;
      if (m < m_end && c0 + s_cq[i] < p.Cin) {
        if (IDENT) {
          src = p.x + (long)m * p.ldx + c0 + s_cq[i];
        } else {
          const int ox = m % p.OW;
          const int q = m / p.OW;
          const int oy = q % p.OH, b = q / p.OH;
          const int sy = oy * p.stride - p.pad + ky * p.dil, sx = ox * p.stride - p.pad + kx * p.dil;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W)
            src = p.x + ((long)(b * p.H + sy) * p.W + sx) * p.ldx + c0 + s_cq[i];
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + T_FLOATS + (wave * 2 + i) * 256),
                                       16, 0, 0);
    }
    ld_m += BKD;
    if (++ld_buf == NST) ld_buf = 0;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int h = lane >> 5;
  const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31);
  const int nsteps = (m_end - m_begin + BKD - 1) / BKD;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nsteps) issue();
  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NST - 1 < nsteps) issue();
    const float* As = smem + cur * ST_FLOATS;
    const float* Bs = As + T_FLOATS;
    if (do_bias && tid < 64) {
#pragma unroll
      for (int k = 0; k < BKD; ++k) bsum += As[k * 64 + tid];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float af = As[(16 * h + i) * 64 + arow], bf = Bs[(16 * h + i) * 64 + bcol];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
    }
    if (++cur == NST) cur = 0;
  }
  const long T = (long)p.kh * p.kw;
  float* slab = p.slab + ((long)split * T + t) * p.Cout * p.Cin;
  const int c = c0 + bcol;
  if (c < p.Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < p.Cout) slab[(long)n * p.Cin + c] = acc[r];
    }
  }
  if (do_bias && tid < 64 && n0 + tid < p.Cout) p.bslab[(long)split * p.Cout + n0 + tid] = bsum;
}

// Weight gradient with fp32 products as six bf16 x bf16 products (see vr_split3): (64 TN) x (64 TC) tile of dW, 2 x 2
// waves with TN x TC accumulators each, 16 contraction rows (one k16 step) per stage, operand tiles [16 rows m][channels]
// by LDS-DMA into a 3-stage ring; the fragments (8 consecutive rows of one channel) are columns of the staged tiles, read
// with conflict-free ds_read_b32 (lanes = consecutive channels) and split in registers.  Same slabs / reduce pass as the
// other weight-gradient kernels.
template <int TN, int TC, bool IDENT, int PROD>      // PROD: 6 = x6, 1 = bf16-rounded operands (one product)
__global__ __launch_bounds__(256, 3) void wgrad_x6_kernel(const WgradArgs p) {
  constexpr int BKD = 16, NST = 3, BN = 64 * TN, BC = 64 * TC;
  constexpr int Y_FLOATS = BKD * BN, X_FLOATS = BKD * BC, ST_FLOATS = Y_FLOATS + X_FLOATS;
  constexpr int NY = Y_FLOATS / 1024, NX = X_FLOATS / 1024, NP = NY + NX;
  static_assert(NY >= 1 && NX >= 1 && NP <= 4, "1 KB DMA pieces per wave");
  __shared__ __attribute__((aligned(16))) float smem[NST * ST_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = blockIdx.x, m_begin, m_end, split;
  if (p.xcd_group) {
    if (!wgrad_rows_xcd(p, bid, m_begin, m_end, split)) return;
  } else {
    wgrad_rows(p, m_begin, m_end, split);
  }
  const int ct = bid % p.c_tiles; bid /= p.c_tiles;
  const int nt = bid % p.n_tiles; bid /= p.n_tiles;
  const int t = bid;
  const int ky = t / p.kw, kx = t - ky * p.kw;
  const int n0 = nt * BN, c0 = ct * BC;
  const bool do_bias = p.bslab != nullptr && ct == 0 && t == 0;
  float bsum = 0.f;
  int y_kr[NY], y_cq[NY], x_kr[NX], x_cq[NX];
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    const int sl = (wave * NY + i) * 64 + lane;
    y_kr[i] = sl / (BN / 4);
    y_cq[i] = 4 * (sl % (BN / 4));
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int sl = (wave * NX + i) * 64 + lane;
    x_kr[i] = sl / (BC / 4);
    x_cq[i] = 4 * (sl % (BC / 4));
  }
  const float* zero_page = vr_zero_page;          // taken once and opaque: see igemm_dma_kernel
  asm volatile("" : "+s"(zero_page));
  int ld_m = m_begin, ld_buf = 0;
  auto issue = [&]() {
    float* stage = smem + ld_buf * ST_FLOATS;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int m = ld_m + y_kr[i];
      const float* src = (m < m_end && n0 + y_cq[i] < p.Cout) ? p.dy + (long)m * p.lddy + n0 + y_cq[i] : zero_page;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + (wave * NY + i) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = ld_m + x_kr[i];
      const float* src = zero_page;
      if (m < m_end && c0 + x_cq[i] < p.Cin) {
        if (IDENT) {
          src = p.x + (long)m * p.ldx + c0 + x_cq[i];
        } else {
          const int ox = m % p.OW;
          const int q = m / p.OW;
          const int oy = q % p.OH, b = q / p.OH;
          const int sy = oy * p.stride - p.pad + ky * p.dil, sx = ox * p.stride - p.pad + kx * p.dil;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W)
            src = p.x + ((long)(b * p.H + sy) * p.W + sx) * p.ldx + c0 + x_cq[i];
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + Y_FLOATS + (wave * NX + i) * 256),
                                       16, 0, 0);
    }
    ld_m += BKD;
    if (++ld_buf == NST) ld_buf = 0;
  };

  f32x16 acc[TN][TC];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int h = lane >> 5;
  const int arow = wm * 32 * TN + (lane & 31), bcol = wn * 32 * TC + (lane & 31);
  const int nsteps = (m_end - m_begin + BKD - 1) / BKD;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nsteps) issue();
  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) {
      if (NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NST - 1 < nsteps) issue();
    const float* Ys = smem + cur * ST_FLOATS;
    const float* Xs = Ys + Y_FLOATS;
    if (do_bias && tid < BN) {
#pragma unroll
      for (int k = 0; k < BKD; ++k) bsum += Ys[k * BN + tid];
    }
    constexpr int NPL = PROD == 6 ? 3 : 1;
    vr_bf16x8 a3[TN][NPL], b3[TC][NPL];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      f32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = Ys[(8 * h + e) * BN + arow + 32 * i];
        hi[e] = Ys[(8 * h + 4 + e) * BN + arow + 32 * i];
      }
      if constexpr (PROD == 6) vr_split3(lo, hi, a3[i]);
      else a3[i][0] = vr_round8(lo, hi);
    }
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      f32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = Xs[(8 * h + e) * BC + bcol + 32 * j];
        hi[e] = Xs[(8 * h + 4 + e) * BC + bcol + 32 * j];
      }
      if constexpr (PROD == 6) vr_split3(lo, hi, b3[j]);
      else b3[j][0] = vr_round8(lo, hi);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        if constexpr (PROD == 6) acc[i][j] = vr_mfma_x6(a3[i], b3[j], acc[i][j]);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][0], acc[i][j], 0, 0, 0);
      }
    if (++cur == NST) cur = 0;
  }
  const long T = (long)p.kh * p.kw;
  float* slab = p.slab + ((long)split * T + t) * p.Cout * p.Cin;
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int c = c0 + bcol + 32 * j;
    if (c >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 32 * TN + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < p.Cout) slab[(long)n * p.Cin + c] = acc[i][j][r];
      }
  }
  if (do_bias && tid < BN && n0 + tid < p.Cout) p.bslab[(long)split * p.Cout + n0 + tid] = bsum;
}

// dw (OIHW) = row_scale[n] * sum_s slab[s][t][n][c] (+ dw);  db[n] = row_scale[n] * sum_s bslab[s][n] (+ db)
// SL lanes share one output (quad): lane sl sums slabs sl, sl + SL, ...; the SL partials are then added in lane
// order through LDS (fixed order: deterministic).  Small weight matrices are split over up to 256 row ranges, and
// a one-thread-per-output loop over that many slabs is a serial chain of dependent-latency loads.
template <int VEC, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slab, const float* bslab, const float* row_scale,
                                                           float* dw, float* db, int S, int T, int Cout, int Cin,
                                                           int accumulate, const float* row_scale2, float* dw2, float* db2,
                                                           const float* w_ls, const float* w_ls2, float* ls_part) {
  constexpr int OUTS = 256 / SL;
  __shared__ float red[SL][OUTS][VEC];
  const long per = (long)T * Cout * Cin;
  // ls_part (layer-scale gradient, T == 1): [stream][per / VEC + Cout] -- per output quad the dot of the RAW weight
  // gradient with the weights, per output channel the raw bias gradient; summed per row by wgrad_rowdot_kernel
  if (blockIdx.y) {      // second stream of a two-stream launch: its own S slabs, outputs and row scale
    slab += (long)S * per;
    if (bslab) bslab += (long)S * Cout;
    row_scale = row_scale2; dw = dw2; db = db2;
    w_ls = w_ls2;
    if (ls_part) ls_part += per / VEC + Cout;
  }
  const long nq = per / VEC;
  const int o = threadIdx.x % OUTS, sl = threadIdx.x / OUTS;
  const long e = (long)blockIdx.x * OUTS + o;
  const long idx = e * VEC;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  if (e < nq) {
#pragma unroll 4
    for (int k = sl; k < S; k += SL) {
      if (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (long)k * per + idx);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += v[j];
      } else {
        s[0] += slab[(long)k * per + idx];
      }
    }
  } else if (db && e < nq + Cout) {
    for (int k = sl; k < S; k += SL) s[0] += bslab[(long)k * Cout + (e - nq)];
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[sl][o][j] = s[j];
  __syncthreads();
  if (sl != 0) return;
#pragma unroll
  for (int k = 1; k < SL; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] += red[k][o][j];
  if (e < nq) {
    const int c = idx % Cin;
    const long q = idx / Cin;
    const int n = q % Cout;
    const int t = q / Cout;
    const float rs = row_scale ? row_scale[n] : 1.f;
    if (ls_part) {
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) dsum += s[j] * w_ls[idx + j];
      ls_part[e] = dsum;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float* d = dw + ((long)n * Cin + c + j) * T + t;
      const float v = s[j] * rs;
      *d = accumulate ? *d + v : v;
    }
  } else if (db && e < nq + Cout) {
    const int n = e - nq;
    float v = s[0];
    if (ls_part) ls_part[e] = v;
    if (row_scale) v *= row_scale[n];
    db[n] = accumulate ? db[n] + v : v;
  }
}

// Layer-scale gradient from the weight gradient of the branch's last 1x1 conv (vr_coc.py:266-271: x + ls * t,
// t = W h + b):  dls[n] = sum_m dy[m,n] t[m,n] = sum_c W[n,c] * (sum_m dy[m,n] h[m,c]) + b[n] * sum_m dy[m,n]
//                       = rowdot(W[n,:], dW_raw[n,:]) + b[n] * db_raw[n]
// -- the branch output t never has to be stored for the backward pass and no pass over (dy, t) is needed.
// wgrad_reduce_kernel leaves per-quad dot partials and the raw bias gradient in ls_part; one wave per output
// channel adds them in a fixed order.
__global__ __launch_bounds__(64) void wgrad_rowdot_kernel(const float* ls_part, const float* bias, float* dls, int Cout,
                                                          int quads_per_row, long stream_stride, int accumulate,
                                                          const float* bias2, float* dls2) {
  if (blockIdx.y) {
    ls_part += stream_stride;
    bias = bias2; dls = dls2;
  }
  const int n = blockIdx.x;
  const float* row = ls_part + (long)n * quads_per_row;
  double acc = 0.0;
  for (int q = threadIdx.x; q < quads_per_row; q += 64) acc += (double)row[q];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    if (bias) acc += (double)bias[n] * (double)ls_part[(long)Cout * quads_per_row + n];
    dls[n] = (accumulate ? dls[n] : 0.f) + (float)acc;
  }
}

// Slab reduction AND layer-scale gradient in one launch (round 5; T == 1, Cin % 4 == 0): one workgroup per output row n, SL
// lanes per quad of the row (lane sl adds slabs sl, sl + SL, ...; the SL partials meet through LDS in lane order), rows longer
// than 256 / SL quads in passes.  The row's dot with the weights -- what wgrad_rowdot_kernel summed from per-quad partials in a
// second launch -- is finished here: per-thread fp64 partials over the thread's quads, then wave and workgroup sums in a fixed
// order.  Same bits run to run; -54 launches per step and no ls_part round trip.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* slab, const float* bslab, const float* row_scale,
                                                                float* dw, float* db, int S, int Cout, int Cin, int accumulate,
                                                                const float* row_scale2, float* dw2, float* db2,
                                                                const float* w_ls, const float* w_ls2, const float* bias,
                                                                const float* bias2, float* dls, float* dls2) {
  constexpr int OUTS = 256 / SL;
  __shared__ float red[SL][OUTS][4];
  __shared__ double wsum[4];
  __shared__ float bsum[16];
  const long per = (long)Cout * Cin;
  if (blockIdx.y) {      // second stream of a two-stream launch
    slab += (long)S * per;
    if (bslab) bslab += (long)S * Cout;
    row_scale = row_scale2; dw = dw2; db = db2; w_ls = w_ls2; bias = bias2; dls = dls2;
  }
  const int n = blockIdx.x, Q = Cin >> 2;
  const int o = threadIdx.x % OUTS, sl = threadIdx.x / OUTS;
  const float rs = row_scale ? row_scale[n] : 1.f;
  const float* row = slab + (long)n * Cin;
  double dot = 0.0;
  for (int q0 = 0; q0 < Q; q0 += OUTS) {
    const int q = q0 + o;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (q < Q) {
#pragma unroll 4
      for (int k = sl; k < S; k += SL) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + (long)k * per + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += v[j];
      }
    }
    if (SL > 1) {
      if (q0) __syncthreads();      // the previous pass's partials have been consumed
#pragma unroll
      for (int j = 0; j < 4; ++j) red[sl][o][j] = s[j];
      __syncthreads();
    }
    if (sl == 0 && q < Q) {
#pragma unroll
      for (int k = 1; k < SL; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += red[k][o][j];
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w_ls + (long)n * Cin + q * 4);
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) dsum += s[j] * wv[j];
      dot += (double)dsum;
      float* d = dw + (long)n * Cin + q * 4;
      f32x4 outv;
#pragma unroll
      for (int j = 0; j < 4; ++j) outv[j] = s[j] * rs;
      if (accumulate) {
        const f32x4 old = *reinterpret_cast<const f32x4*>(d);
#pragma unroll
        for (int j = 0; j < 4; ++j) outv[j] = old[j] + outv[j];
      }
      *reinterpret_cast<f32x4*>(d) = outv;
    }
  }
  // raw bias gradient of the row: 16 lanes over the slabs, added in lane order
  if (bslab && threadIdx.x < 16) {
    float b = 0.f;
    for (int k = threadIdx.x; k < S; k += 16) b += bslab[(long)k * Cout + n];
    bsum[threadIdx.x] = b;
  }
  dot = wave_sum(dot);               // (threads with sl != 0 hold 0)
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    float braw = 0.f;
    if (bslab) {
#pragma unroll
      for (int k = 0; k < 16; ++k) braw += bsum[k];
      if (db) {
        const float v = row_scale ? braw * rs : braw;
        db[n] = accumulate ? db[n] + v : v;
      }
    }
    if (bias) acc += (double)bias[n] * (double)braw;
    dls[n] = (accumulate ? dls[n] : 0.f) + (float)acc;
  }
}

__global__ void pack_weight_kernel(const float* w, float* out, int Cout, int Cin, int T) {
  const long total = (long)T * Cout * Cin;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = e % Cin;
  const long q = e / Cin;
  const int n = q % Cout;
  const int t = q / Cout;
  out[e] = w[((long)n * Cin + c) * T + t];
}

}  // namespace

// igemm_bf16.hip
int vr_igemm_bf16_launch(const void* args, int mode, hipStream_t st);
// igemm_planes.hip: x6 with pre-split weights, A fragments from global memory; returns 1 when the shape has no such kernel
int vr_igemm_planes_reg_launch(const void* args, const void* planes, int variant, long M, int S, hipStream_t st);
// igemm_planes.hip: K = 64 / 128 over big maps, B resident in LDS, barrier-free A streams; returns 1 when the shape has no such kernel
int vr_igemm_planes_stream_launch(const void* args, const void* planes, long M, hipStream_t st);
int vr_wgrad_bf16_launch(const void* args, int ident, int blocks_x, int splits, int streams, hipStream_t st);

// narrowconv.hip
bool vr_narrow_conv_ok(int Cin, int Cout, int kh, int kw, int stride, int pad);
bool vr_narrow_wgrad_ok(int Cin, int Cout, int kh, int kw, int stride, int pad);
int vr_narrow_conv(int mode, const float* a, long lda, const float* w, const float* bias, float* y, long ldy, long M, long HW,
                   int Cin, int Cout, int out_nchw, int out_ctot, int out_coff, int accumulate, hipStream_t st);
long vr_narrow_wgrad_workspace(long M, int Cin, int Cout);
int vr_narrow_wgrad(const float* x, long ldx, const float* dy, long lddy, long M, int Cin, int Cout, void* workspace,
                    int want_bias, float** slab_out, float** bslab_out, int* splits, hipStream_t st);

// tinyconv.hip
int vr_tiny_conv(int mode, const float* a, long lda, const float* w, const float* bias, float* y, long ldy, int B, int H,
                 int W, int Cin, int Cout, int k, int pad, int dil, int accumulate, hipStream_t st);
long vr_tiny_wgrad_workspace(long npix, int Cin, int Cout, int T);
int vr_tiny_wgrad(const float* x, long ldx, const float* dy, long lddy, float* dw, float* db, const float* row_scale,
                  int B, int H, int W, int Cin, int Cout, int k, int pad, int dil, int accumulate, void* workspace,
                  hipStream_t st);
int vr_patch_dgrad(const float* dy, long lddy, const float* w, float* dx, long lddx, int B, int H, int W, int Cin,
                   int Cout, int k, int accumulate, hipStream_t st);

static bool tiny_shape(int H, int W, int Cin, int OH, int OW, int Cout, int kh, int kw, int stride) {
  return Cin <= 8 && Cout <= 8 && stride == 1 && kh == kw && (kh == 1 || kh == 3) && OH == H && OW == W;
}

// Tile of the LDS-DMA x6 / bf16 kernels for a GEMM of M rows and CN columns: 22 = 128 x 128, 21 = 128 x 64, 0 = none
// (too few tiles to fill the chip, or <= 32 columns).
static int vr_dma_tile(long M, int CN) {
  static const int min_n = vr_tune("VRNET_X6_MIN_N", 96);
  static const int min_tiles = vr_tune("VRNET_X6_MIN_TILES", 256);
  static const int force = vr_tune("VRNET_X6_TILE", 0);     // tuning aid: 22 / 21
  const long mt = vr_cdiv(M, 128), nt22 = vr_cdiv(CN, 128), nt21 = vr_cdiv(CN, 64);
  const bool waste22 = nt22 * 128 - CN > 16 * nt22;          // more than 12 % of the column tiles is padding
  int tile = 0;
  if (CN >= min_n && mt * nt22 >= 2 * min_tiles && !waste22) tile = 22;
  else if (CN > 32 && mt * nt21 >= min_tiles) tile = 21;
  else if (CN >= min_n && mt * nt22 >= min_tiles) tile = 22;
  if (force && tile) tile = force;
  return tile;
}

// Split contraction (round 4) for the layers the rule above turns away for lack of tiles (16 x 16 maps: 2048 rows x 256-512
// columns = 64-128 tiles of 128 x 64 on 256 CUs, which then ran on 64 x 64 fp32-MFMA tiles at 40-70 TFLOP/s): `S` workgroups
// per tile each take 1/S of the K loop and leave raw accumulators in a slab, igemm_splitk_finish_kernel adds the slabs in
// order and runs the epilogue.  Returns the tile (21) and S, or the unsplit tile of vr_dma_tile with S = 1, or 0.
static int vr_dma_plan(long M, int CN, long ktot, long ws_bytes, int* S_out) {
  *S_out = 1;
  const int tile = vr_dma_tile(M, CN);
  if (tile || CN <= 32 || ws_bytes <= 0) return tile;
  static const int on = vr_tune("VRNET_SPLITK", 1);
  static const int target = vr_tune("VRNET_SPLITK_TARGET", 512);         // workgroups wanted (2 per CU)
  // K16 steps per split at least (measured in the step: with 8 the 2 048 x 512 x 512 layers went 21.8 -> 30 us per call --
  // the finishing launch costs more than eight-step splits save; 2 048 x 512 x 2 048: 61.6 -> 46.8 us)
  static const int min_steps = vr_tune("VRNET_SPLITK_MIN_STEPS", 32);
  if (!on) return 0;
  const long mt = vr_cdiv(M, 128), nt21 = vr_cdiv(CN, 64), tiles = mt * nt21, steps = ktot / 16;
  long S = vr_cdiv(target, tiles);
  if (S > steps / min_steps) S = steps / min_steps;
  if (S > 8) S = 8;
  if (S < 2 || tiles * S < 192) return 0;
  const long need = S * (8 * vr_cdiv(mt, 8) * nt21) * 8192 * 4;
  if (need > ws_bytes) return 0;
  *S_out = (int)S;
  return 21;
}

/* Tile and split count vrnet_conv2d_f32 uses at precision 2 / 3 when it is given a workspace (ktot = contraction length
 * Cin * kh * kw resp. Cout * kh * kw): the unsplit tile of vrnet_conv2d_dma_tile with *splits = 1, or 21 with *splits >= 2
 * for the small maps; vrnet_conv2d_splitk_workspace: the bytes that split needs (0: none). */
extern "C" int vrnet_conv2d_dma_plan(long rows, int cols, long ktot, int* splits) {
  int S = 1;
  const int tile = vr_dma_plan(rows, cols, ktot, 1L << 40, &S);
  if (splits) *splits = S;
  return tile;
}

extern "C" long vrnet_conv2d_splitk_workspace(long rows, int cols, long ktot) {
  int S = 1;
  if (vr_dma_plan(rows, cols, ktot, 1L << 40, &S) == 0 || S < 2) return 0;
  return (long)S * (8 * vr_cdiv(vr_cdiv(rows, 128), 8) * vr_cdiv(cols, 64)) * 8192 * 4;
}

/* Which LDS-DMA tile vrnet_conv2d_f32 would use at precision 2 / 3 for B*MH*MW GEMM rows and CN GEMM columns (mode 0:
 * output pixels x Cout; mode 1: input pixels x Cin): 22, 21 or 0 = none (precision 3 is then rejected, precision 2 falls
 * back to the fp32 MFMA).  Alignment requirements (16-byte rows, channel counts % 4) are the caller's, as for precision 1. */
extern "C" int vrnet_conv2d_dma_tile(long rows, int cols) { return vr_dma_tile(rows, cols); }

extern "C" int vrnet_conv2d_f32(const float* a, long lda, const float* w, const float* bias, float* y, long ldy,
                                int B, int H, int W, int Cin, int OH, int OW, int Cout, int kh, int kw, int stride,
                                int pad, int dil, int mode, int act, float* ypre, long ldypre, const float* res,
                                long ldres, const float* res_scale, const float* kscale, const float* aux,
                                long ldaux, int out_nchw, int out_ctot, int out_coff, int accumulate,
                                double* stats, int precision, int pair_rows, const float* w2, const float* bias2,
                                const float* res_scale2, const float* kscale2, const void* w_planes,
                                const vrnet_conv_colstats* colstats, void* workspace, long workspace_bytes, void* stream) {
  VR_CHECK_ARG(a && w && y, "conv2d: null tensor");
  if (vr_ablated("igemm")) return VR_OK;
  {   // finer timing ablations by output-row class (diagnostic): stage-0/1 maps, stage-2 maps, neck / head maps
    const long rows_ = (long)B * (mode == 0 ? OH * OW : H * W);
    if (rows_ >= 32768 ? vr_ablated("igemm_big") : (rows_ > 2048 ? vr_ablated("igemm_mid") : vr_ablated("igemm_small"))) return VR_OK;
  }
  VR_CHECK_ARG(pair_rows >= 0 && (pair_rows == 0 || (w2 && pair_rows % 128 == 0 && (!bias == !bias2) &&
                                                     (!res_scale == !res_scale2) && (!kscale == !kscale2))),
               "conv2d: a two-stream launch needs the second parameter set and a first-stream row count that is a "
               "multiple of 128");
  VR_CHECK_ARG(precision >= 0 && precision <= 3, "conv2d: precision 0 (fp32 MFMA), 1 / 3 (bf16 operands, fp32 accumulate) "
                                                 "or 2 (fp32 products as six bf16 x bf16 products, fp32 accumulate)");
  VR_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && OH > 0 && OW > 0 && Cout > 0, "conv2d: bad shape");
  VR_CHECK_ARG(kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d: bad geometry");
  VR_CHECK_ARG((H + 2 * pad - dil * (kh - 1) - 1) / stride + 1 == OH &&
                   (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1 == OW,
               "conv2d: output size %dx%d inconsistent with input %dx%d k%d s%d p%d d%d", OH, OW, H, W, kh,
               stride, pad, dil);
  const bool plain = !ypre && !res && !kscale && !aux && !out_nchw && act == 0 && !stats && precision != 1 && !pair_rows &&
                     !colstats;
  if (plain && tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride)) vr_note_kernel(4);
  if (plain && tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride))
    return vr_tiny_conv(mode, a, lda, w, mode == 0 ? bias : nullptr, y, ldy, B, H, W, Cin, Cout, kh, pad, dil, accumulate,
                        vr_stream(stream));
  // narrow outputs over wide inputs (head predictions, seg logits): direct HBM-streaming kernels (narrowconv.hip)
  if ((vr_tune("VRNET_NARROW", 3) >> mode & 1) && !colstats && !ypre && !res && !kscale && !aux && act == 0 && !stats && !pair_rows && precision != 1 &&
      precision != 3 && vr_narrow_conv_ok(Cin, Cout, kh, kw, stride, pad) && !tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride) && vr_aligned16(w) &&
      (mode == 0 ? (lda % 4 == 0 && vr_aligned16(a)) : (!out_nchw && ldy % 4 == 0 && vr_aligned16(y)))) {
    VR_CHECK_ARG(lda >= (mode == 0 ? Cin : Cout) && (out_nchw || ldy >= (mode == 0 ? Cout : Cin)),
                 "conv2d: row stride smaller than channel count");
    vr_note_kernel(5);
    return vr_narrow_conv(mode, a, lda, w, mode == 0 ? bias : nullptr, y, ldy, (long)B * H * W, (long)H * W, Cin, Cout,
                          mode == 0 ? out_nchw : 0, out_ctot, out_coff, accumulate, vr_stream(stream));
  }
  if (plain && mode == 1 && Cin <= 8 && kh == kw && kh == stride && pad == 0 && dil == 1 && H == OH * kh &&
      W == OW * kw && (size_t)kh * kw * Cout * Cin * 4 <= 60000) {
    vr_note_kernel(4);
    return vr_patch_dgrad(a, lda, w, y, ldy, B, H, W, Cin, Cout, kh, accumulate, vr_stream(stream));
  }
  IgemmArgs p{};
  p.a = a; p.lda = lda; p.w = w; p.bias = bias; p.y = y; p.ldy = ldy;
  p.ypre = ypre; p.ldypre = ldypre; p.res = res; p.ldres = ldres; p.res_scale = res_scale;
  p.kscale = kscale; p.aux = aux; p.ldaux = ldaux;
  p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.dil = dil;
  p.mode = mode; p.act = act; p.out_nchw = out_nchw; p.out_ctot = out_ctot; p.out_coff = out_coff;
  p.accumulate = accumulate; p.wtap = (long)Cout * Cin; p.Cin = Cin;
  if (mode == 0) {
    p.MH = OH; p.MW = OW; p.SH = H; p.SW = W; p.CK = Cin; p.CN = Cout;
  } else {
    p.MH = H; p.MW = W; p.SH = OH; p.SW = OW; p.CK = Cout; p.CN = Cin;
  }
  const long M = (long)B * p.MH * p.MW;
  VR_CHECK_ARG(M < (1L << 31) && (long)B * p.SH * p.SW < (1L << 31), "conv2d: too many pixels");
  p.M = (int)M;
  p.perm2 = (mode == 1 && stride == 2 && p.MH % 2 == 0 && p.MW % 2 == 0 && (M / 4) % 128 == 0) ? 1 : 0;
  VR_CHECK_ARG(lda >= p.CK && (out_nchw || ldy >= p.CN), "conv2d: row stride smaller than channel count");
  p.a_vec = (p.CK % 4 == 0) && (lda % 4 == 0) && vr_aligned16(a);
  p.b_vec = (Cin % 4 == 0) && vr_aligned16(w);
  p.e_vec = !out_nchw && (p.CN % 4 == 0) && (ldy % 4 == 0) && vr_aligned16(y) &&
            (!bias || vr_aligned16(bias)) && (!ypre || ((ldypre % 4 == 0) && vr_aligned16(ypre))) &&
            (!res || ((ldres % 4 == 0) && vr_aligned16(res))) && (!res_scale || vr_aligned16(res_scale)) &&
            (!aux || ((ldaux % 4 == 0) && vr_aligned16(aux)));
  p.stats = stats;
  p.stats_nb = (int)vr_cdiv(p.CN, 32);
  p.dbg_fake_presplit = vr_tune("VRNET_X6_FAKE_PRESPLIT", 0);
  if (colstats) {
    VR_CHECK_ARG(colstats->partial && (!colstats->tile_totals || colstats->gamma), "conv2d: column statistics need `partial` (and gamma "
                                                                                 "with tile_totals)");
    p.col_part = colstats->partial; p.col_x2 = colstats->x2; p.ld_col_x2 = colstats->ldx2;
    p.col_gamma = colstats->gamma; p.col_tot = colstats->tile_totals;
  }
  p.pair_rows = pair_rows; p.w2 = w2; p.bias2 = bias2; p.res_scale2 = res_scale2; p.kscale2 = kscale2;
  VR_CHECK_ARG(!pair_rows || (pair_rows < M && 2L * pair_rows == M), "conv2d: the two streams must have equal row counts");
  if (pair_rows && p.perm2) {
    if ((M / 8) % 128 == 0) p.pair_rows = (int)(M / 8);      // parity-major rows: the streams split every parity class
    else p.perm2 = 0;
  }
  if (pair_rows) {
    p.b_vec = p.b_vec && vr_aligned16(w2);
    p.e_vec = p.e_vec && (!bias2 || vr_aligned16(bias2)) && (!res_scale2 || vr_aligned16(res_scale2));
  }
  VR_CHECK_ARG(!colstats || (p.e_vec && !p.perm2 && p.CN > 32 && (!colstats->x2 || (colstats->ldx2 % 4 == 0 && vr_aligned16(colstats->x2)))),
               "conv2d: column statistics need the vector epilogue (NHWC, channel counts %% 4 == 0, 16-byte rows), > 32 output "
               "channels and no stride-2 data gradient");
#ifndef VR_IGEMM_STAMP2
  VR_CHECK_ARG(!stats || (p.e_vec && !p.perm2 && mode == 0 && ((long)p.MH * p.MW) % 32 == 0 && p.CN > 32),
               "conv2d: output statistics need the vector epilogue, a forward conv, > 32 output channels and a map of a "
               "multiple of 32 pixels");
#endif
  dim3 block(256);
  hipStream_t st = vr_stream(stream);
  // bf16-rounded operands: the forward launches whose grid fills the chip run on the LDS-DMA tile kernels below (as do
  // data gradients at precision 3, standard weight layout); everything else on the register-staged bf16 kernel
  const bool bf16_tile_fwd = precision == 1 && mode == 0 && p.a_vec && p.b_vec && p.CN > 32 && vr_dma_tile(M, p.CN) != 0 &&
                             (!pair_rows || pair_rows % 128 == 0);
  if (precision == 1 && !bf16_tile_fwd) {
    // bf16 operands: both operands must be contraction-contiguous 16-byte rows; in mode 1 `w` is the TRANSPOSED pack
    // (vrnet_pack_weight_t_f32, which also folds kscale in), so kscale must not be passed again
    VR_CHECK_ARG(p.a_vec && p.CK % 4 == 0 && vr_aligned16(w) && p.CN > 32 && !kscale,
                 "conv2d: the bf16 path needs 16-byte aligned rows, a contraction that is a multiple of 4, more than 32 "
                 "output channels and (mode 1) the transposed weight pack with kscale folded in");
    p.perm2 = 0;
    p.pair_rows = pair_rows;
    vr_igemm_bf16_launch(&p, mode, st);
    vr_note_kernel(3);
    VR_LAUNCH_CHECK("conv2d(bf16)");
    return VR_OK;
  }
  // Tile choice: 128-row tiles while they fill the chip (256 CUs x >= 2 workgroups); otherwise 64 x 64 tiles,
  // which quadruple the workgroup count of the small-M layers (M = 2048 / 8192 at the 16x16 / 32x32 stages).
  const long mt128 = vr_cdiv(M, 128);
  const int bn128 = p.CN > 64 ? 128 : (p.CN > 32 ? 64 : 32);
  const long blocks128 = mt128 * vr_cdiv(p.CN, bn128);
  const bool vec = p.a_vec && p.b_vec;
  static const int force_cfg = vr_tune("VRNET_IGEMM_CFG", -1);   // tuning aid
  static const int force_bk = vr_tune("VRNET_IGEMM_BK", 0);   // tuning aid
  const bool bk32 = force_bk == 32;        // BK = 32 measured 3 % slower over the net's shapes (tools/tune_igemm.py --bk)
#define VR_IGEMM_(BM_, BN_, BK_, TM_, TN_, WM_, WN_, GRID)                                                              \
  do {                                                                                                                  \
    if (mode == 0) {                                                                                                    \
      if (vec) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, BK_, TM_, TN_, WM_, WN_, 0, true>), GRID, block, 0, st, p);   \
      else hipLaunchKernelGGL((igemm_kernel<BM_, BN_, BK_, TM_, TN_, WM_, WN_, 0, false>), GRID, block, 0, st, p);      \
    } else {                                                                                                            \
      if (vec) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, BK_, TM_, TN_, WM_, WN_, 1, true>), GRID, block, 0, st, p);   \
      else hipLaunchKernelGGL((igemm_kernel<BM_, BN_, BK_, TM_, TN_, WM_, WN_, 1, false>), GRID, block, 0, st, p);      \
    }                                                                                                                   \
  } while (0)
#define VR_IGEMM(BM_, BN_, TM_, TN_, WM_, WN_, GRID)                     \
  do {                                                                   \
    if (bk32) VR_IGEMM_(BM_, BN_, 32, TM_, TN_, WM_, WN_, GRID);         \
    else VR_IGEMM_(BM_, BN_, 16, TM_, TN_, WM_, WN_, GRID);              \
  } while (0)
  // Measured on MI355X over every conv shape of the net (tools/tune_igemm.py, phi = l, bs 8, 512 px): 64 x 64 tiles
  // (7 workgroups per CU) beat 128 x 128 (2 per CU) and 128 x 64 on 90 % of the shapes -- 21.4 vs 30.7 / 25.9 ms
  // per step over all forward + data-gradient launches; the exceptions are within 10 %.
  (void)blocks128;
  static const int use_dma = vr_tune("VRNET_IGEMM_DMA", 1);   // tuning aid
  // Measured per shape (bench.py --detail): the DMA ring wins where the grid cannot fill the chip with 8 workgroups
  // per CU (M <= 8192 pixels: +5..40 %) and on long contractions; the register-staged kernel keeps the large-M,
  // short-K layers (2048 row tiles x few K steps: its 8 workgroups per CU hide the store-heavy epilogues better).
  const long ktot = (long)p.CK * kh * kw;
  const bool dma_ok = use_dma && force_cfg < 0 && vec && p.CN > 32 && (!kscale || p.CK <= 1024);
  // 64 x 64 ring tiles where the grid cannot fill the chip with 8 workgroups per CU (M <= 8192 pixels: +5..40 %) and
  // on long contractions; the register-staged kernel keeps the large-M, short-K layers (2048 row tiles x few K steps:
  // its 8 workgroups per CU hide the store-heavy epilogues better).  128 x 128 ring tiles (T = 2) were measured on the
  // large-M layers with >= 768 such tiles: 4-11 % slower than either (K <= 256 there: prologue / epilogue bound), so
  // they are not dispatched.  All measured with bench.py --detail.
  const bool dma_shape = use_dma == 2 || M <= 8192 || (M <= 32768 && ktot >= 1024);
  // precision 2 (x6): 128 x 128 x 16 ring tiles with 2 x 2 accumulators per wave -- each split fragment feeds two
  // MFMA groups, which is what pays for the split (on 64 x 64 tiles the conversions cost what the faster MFMA saves)
  const bool bf16_tiles = precision == 3 || (precision == 1 && mode == 0);   // (mode 1 at precision 1 brings the transposed pack)
  if ((precision == 2 || bf16_tiles) && dma_ok && (!pair_rows || p.pair_rows % 128 == 0)) {
    const long mt = vr_cdiv(M, 128), nt22 = vr_cdiv(p.CN, 128), nt21 = vr_cdiv(p.CN, 64);
    int S = 1;
    // (the bf16-rounded tiles of precision 1 / 3 keep the unsplit rule: their callers asked vrnet_conv2d_dma_tile)
    // contraction length that counts for the split: only taps that reach the image for at least one row (a 3 x 3 conv with
    // dilation 18 on a 16 x 16 map is its centre tap)
    long k_live = ktot;
    if (kh * kw > 1) {
      auto live = [&](int k, int n_m, int n_s) {
        int c = 0;
        for (int t = 0; t < k; ++t) {
          bool any = false;
          for (int o = 0; o < n_m && !any; ++o) {
            if (mode == 0) {
              const int sidx = o * stride - pad + t * dil;
              any = sidx >= 0 && sidx < n_s;
            } else {
              const int tt = o + pad - t * dil;
              any = tt >= 0 && tt % stride == 0 && tt / stride < n_s;
            }
          }
          c += any;
        }
        return c;
      };
      k_live = (long)p.CK * live(kh, p.MH, p.SH) * live(kw, p.MW, p.SW);
    }
    const int tile = precision == 2 ? vr_dma_plan(M, p.CN, k_live, workspace ? workspace_bytes : 0, &S) : vr_dma_tile(M, p.CN);
    if (S > 1) { p.ksplit = S; p.kslab = reinterpret_cast<float*>(workspace); }
#define VR_TILE_LAUNCH(PR)                                                                                              \
  do {                                                                                                                  \
    if (tile == 22) {                                                                                                   \
      dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * nt22));                                                                 \
      if (mode == 0) hipLaunchKernelGGL((igemm_dma_kernel<0, 3, 2, 2, PR>), grid, block, 0, st, p, (int)mt, (int)nt22); \
      else hipLaunchKernelGGL((igemm_dma_kernel<1, 3, 2, 2, PR>), grid, block, 0, st, p, (int)mt, (int)nt22);           \
    } else {                                                                                                            \
      dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * nt21 * S));                                                             \
      if (mode == 0) hipLaunchKernelGGL((igemm_dma_kernel<0, 3, 2, 1, PR>), grid, block, 0, st, p, (int)mt, (int)nt21); \
      else hipLaunchKernelGGL((igemm_dma_kernel<1, 3, 2, 1, PR>), grid, block, 0, st, p, (int)mt, (int)nt21);           \
      if (S > 1)                                                                                                        \
        hipLaunchKernelGGL((igemm_splitk_finish_kernel<2, 1, 2, 2>), dim3((unsigned)(8 * vr_cdiv(mt, 8) * nt21)), block, 0, st, \
                           p, (int)mt, (int)nt21);                                                                      \
    }                                                                                                                   \
  } while (0)
    if (tile && precision == 2 && w_planes && kh == 1 && kw == 1 && stride == 1 && pad == 0 && !pair_rows && p.CK % 16 == 0 &&
        vr_tune("VRNET_X6_PLANES", 1)) {
      // pre-split weights (vrnet_conv_planes_pack_f32; in mode 1 the pack holds the transposed weights with kscale folded in)
      IgemmArgs q = p;
      q.kscale = nullptr;
      const int JB = (int)(((p.CN + 127) >> 7) << 1);
      // round 6: A fragments straight from global memory (igemm_planes.hip); variant = 100 NW + 10 TN + workgroups per CU.
      // Measured alone, warm / operands rotated through > 256 MB (profiles/r06_planes_reg_probe.txt): the 256 x 64 tile with
      // eight waves on one B stage takes 10-17 % less time where the 128 x 64 grid leaves the chip under-filled and the
      // contraction is long (8 192 rows x 320 columns, K >= 512: 70.5 -> 58.8 us warm, 75.5 -> 68.2 cold); elsewhere the
      // variants are within +-5 % of igemm_planes_kernel cold and the step does not move (24.33 vs 24.26-24.31 ms, same call),
      // so only that class is dispatched (-1 = this rule; the diagnostic build can force a variant for every launch).
      // Round 6: two more forms of this GEMM (igemm_planes.hip), both measured and NEITHER dispatched by the product build (the
      // diagnostic build can force them: VRNET_PLANES_STREAM_MIN_ROWS / VRNET_PLANES_REG21 / _REG22; tests/test_planes_reg.py runs
      // the parity cases on them).  Alone they are faster -- profiles/r06_planes_reg_probe.txt, r06_planes_stream_probe.txt:
      // A fragments from global memory 5-17 % on the under-filled 8 192-row grids with long contractions (70.5 -> 58.8 us on
      // 256 x 64 tiles of eight waves), resident B + barrier-free A streams 5-15 % at 131 072 rows x <= 128 columns -- but with
      // operands rotated through > 256 MB (as inside the step) the gains shrink to 0-10 %, and IN the step they are not there:
      // same call, ms per step: round-5 head 25.09-25.15, the 256 x 64 rule on 25.27-25.38, rule off 25.16-25.26, streaming rule on
      // or off 25.38-25.47 vs 25.39-25.46; every 128 x 64 variant 24.17-24.35 against 24.14-24.50 (profiles/r06_planes_reg_step.txt).
      static const int stream_min = vr_tune("VRNET_PLANES_STREAM_MIN_ROWS", VR_PLANES_STREAM_MIN_ROWS);
      static const int stream_max_n = vr_tune("VRNET_PLANES_STREAM_MAX_COLS", 128);
      if (stream_min > 0 && S == 1 && M >= stream_min && p.CN <= stream_max_n && !p.perm2 &&
          vr_igemm_planes_stream_launch(&q, w_planes, M, st) == 0) {
        vr_note_kernel(9);
        VR_LAUNCH_CHECK("conv2d(x6, pre-split weights, resident B)");
        return VR_OK;
      }
      // variant = 100 NW + 10 TN + workgroups per CU; -1 = 256 x 64 tiles where the 128 x 64 grid under-fills the chip and the
      // contraction is long (the rule that measured +0.2 ms in the step)
      static const int reg21 = vr_tune("VRNET_PLANES_REG21", VR_PLANES_REG21), reg22 = vr_tune("VRNET_PLANES_REG22", VR_PLANES_REG22);
      int reg_variant = tile == 22 ? reg22 : reg21;
      if (reg_variant < 0) reg_variant = (tile == 21 && S == 1 && M <= 8192 && p.CK >= 512 && p.CN <= 512) ? 812 : 0;
      if (reg_variant && vr_igemm_planes_reg_launch(&q, w_planes, reg_variant, M, S, st) == 0) {
        if (S > 1)
          hipLaunchKernelGGL((igemm_splitk_finish_kernel<1, 2, 4, 1>), dim3((unsigned)(8 * vr_cdiv(mt, 8) * nt21)), block, 0, st, q,
                             (int)mt, (int)nt21);
        vr_note_kernel(9);
        VR_LAUNCH_CHECK("conv2d(x6, pre-split weights, A from global memory)");
        return VR_OK;
      }
      if (tile == 22) {
        dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * nt22));
        // 128 x 128 tile: two stages (40 KB, 3 workgroups per CU) measured 0.2 ms per step ahead of three (60 KB, 2 per CU)
        if (vr_tune("VRNET_PLANES_NST2", 1))
          hipLaunchKernelGGL((igemm_planes_kernel<2, 2>), grid, block, 0, st, q, reinterpret_cast<const unsigned char*>(w_planes), JB,
                             (int)mt, (int)nt22);
        else
          hipLaunchKernelGGL((igemm_planes_kernel<2, 3>), grid, block, 0, st, q, reinterpret_cast<const unsigned char*>(w_planes), JB,
                             (int)mt, (int)nt22);
      } else {
        dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * nt21 * S));
        hipLaunchKernelGGL((igemm_planes_kernel<1, 3>), grid, block, 0, st, q, reinterpret_cast<const unsigned char*>(w_planes), JB,
                           (int)mt, (int)nt21);
        if (S > 1)
          hipLaunchKernelGGL((igemm_splitk_finish_kernel<1, 2, 4, 1>), dim3((unsigned)(8 * vr_cdiv(mt, 8) * nt21)), block, 0, st, q,
                             (int)mt, (int)nt21);
      }
      vr_note_kernel(9);
      VR_LAUNCH_CHECK("conv2d(x6, pre-split weights)");
      return VR_OK;
    }
    if (tile) {
      if (precision == 2) VR_TILE_LAUNCH(6);
      else VR_TILE_LAUNCH(1);
      vr_note_kernel(precision == 2 ? 6 : 3);
      VR_LAUNCH_CHECK("conv2d(dma tiles)");
      return VR_OK;
    }
#undef VR_TILE_LAUNCH
  }
  VR_CHECK_ARG(precision != 3, "conv2d: precision 3 (bf16-rounded operands on the LDS-DMA tiles, standard weight layout) has "
                               "no kernel for this shape / alignment: ask vrnet_conv2d_dma_tile first");
  if (dma_ok && dma_shape) {
    const int MT = (int)vr_cdiv(M, 64), NT = (int)vr_cdiv(p.CN, 64);
    dim3 grid((unsigned)(8 * vr_cdiv(MT, 8) * NT));
    // (rings of 4 / 6 stages for launches with <= 2 / 1 workgroups per CU were measured: 3-10 % slower -- the stage
    // time there is not DMA latency but per-stage issue overhead, see the interleaved issue in the kernel)
    if (mode == 0) hipLaunchKernelGGL((igemm_dma_kernel<0, 3, 1, 1>), grid, block, 0, st, p, MT, NT);
    else hipLaunchKernelGGL((igemm_dma_kernel<1, 3, 1, 1>), grid, block, 0, st, p, MT, NT);
    vr_note_kernel(2);
    VR_LAUNCH_CHECK("conv2d");
    return VR_OK;
  }
  int cfg = p.CN > 32 ? 2 : 3;
  static const int narrow = vr_tune("VRNET_IGEMM_NARROW", 1);   // tuning aid
  if (narrow && vec && p.CN > 32 && p.CN <= 192 && ktot >= 512) cfg = 1;      // 128 x 64 tiles for narrow outputs
  if (force_cfg >= 0 && p.CN > 32) cfg = (force_cfg == 0 && p.CN <= 64) ? 1 : force_cfg;
  if (cfg == 2) {
    dim3 grid(vr_cdiv(M, 64), vr_cdiv(p.CN, 64));
    VR_IGEMM(64, 64, 1, 1, 2, 2, grid);
  } else if (cfg == 1) {
    dim3 grid(mt128, vr_cdiv(p.CN, 64));
    VR_IGEMM(128, 64, 2, 1, 2, 2, grid);
  } else if (bn128 == 128) {
    dim3 grid(mt128, vr_cdiv(p.CN, 128));
    VR_IGEMM(128, 128, 2, 2, 2, 2, grid);
  } else if (bn128 == 64) {
    dim3 grid(mt128, 1);
    VR_IGEMM(128, 64, 2, 1, 2, 2, grid);
  } else {
    dim3 grid(mt128, 1);
    VR_IGEMM(128, 32, 1, 1, 4, 1, grid);
  }
#undef VR_IGEMM
#undef VR_IGEMM_
  vr_note_kernel(1);
  VR_LAUNCH_CHECK("conv2d");
  return VR_OK;
}

// Tile + split plan.  cfg 0: 128 x {128,64,32} tiles; cfg 1: 64 x 64 tiles (more workgroups for the
// small-M stages).  Every split costs one slab of |dW| floats written and read back by the reduce pass,
// so splits are bounded by >= 128 contraction rows each and <= 48 MB of slabs.
static void wgrad_plan(long M, int Cin, int Cout, int T, int* cfg, int* bn, int* n_tiles, int* c_tiles, int* S,
                       int* rows, int bf16 = 0) {
  const long wsz = (long)T * Cout * Cin;
  if (bf16 == 2) {      // x6 kernels: cfg = 10 TN + TC, tiles of (64 TN) x (64 TC); 0 = no x6 kernel for this shape
    static const int x6_wgrad = vr_tune("VRNET_X6_WGRAD", 1);   // tuning aid
    *cfg = 0;
    if (!x6_wgrad || Cout <= 32 || Cin <= 32) return;
    const int tn = Cout > 64 ? 2 : 1, tc = Cin > 64 ? 2 : 1;
    if (tn == 1 && tc == 1) return;
    // (128 x 256 / 256 x 128 tiles -- 2 x 4 accumulators per wave, six fragment splits per 48 MFMAs instead of four per 24 --
    // were measured in round 4: 8-45 % SLOWER on every shape of the net alone (8 192 x 320 x 1 280: 63.5 -> 70.2 us) and
    // 0.1-0.2 ms per step slower: at two workgroups per CU the loop loses more to latency than the splits cost.)
    *cfg = 10 * tn + tc; *bn = 64 * tc;
    *n_tiles = (int)vr_cdiv(Cout, 64 * tn); *c_tiles = (int)vr_cdiv(Cin, 64 * tc);
    const long tiles = (long)*n_tiles * *c_tiles * T;
    static const int x6_target = vr_tune("VRNET_X6_WGRAD_TARGET", 768);      // tuning aid: workgroups wanted when tiles > 96
    long s = vr_cdiv(x6_target, tiles);
    long smax = vr_cdiv(M, 256);
    const long sbytes = (48L << 20) / (wsz * 4);
    // XCD-grouped launch (wgrad_rows_xcd): all tiles of a row split on one XCD, 8 k splits per group, `per_xcd` workgroups per XCD.
    // Rounds 2-5 filled an XCD (96 = 32 CUs x 3 workgroups: the fastest launch ALONE).  Round 6, in the step: a third of that.
    // Every split costs a slab of |dW| written and read back (8 192 x 320 x 1 280: 24 splits = 79 MB of slab traffic beside 52 MB
    // of operands), and the weight gradients run beside the chains, not on them: thinner launches leave the chains the CUs and
    // the bytes.  Same call, ms per step, fp32 bs 8 (profiles/r06_wgrad_split_sweep.txt): 96: 25.08-25.15 / 24.73-24.94 on a
    // second box; 64: 24.53-24.97; 48: 24.68-24.84; **32: 24.93-24.98 / 24.37-24.47**; 24: 24.97-25.03; 16: 25.15-25.39; 8: 26.6.
    static const int s8 = vr_tune("VRNET_X6_WGRAD_S8", 1);      // tuning aid
    static const int per_xcd = vr_tune("VRNET_X6_WGRAD_PER_XCD", 32);      // tuning aid
    static const int per_xcd_small = vr_tune("VRNET_X6_WGRAD_PER_XCD_SMALL", 0);      // tuning aid: for <= 16 tiles (0 = per_xcd)
    if (s8 && tiles <= 96) {
      long g = 8 * ((per_xcd_small && tiles <= 16 ? per_xcd_small : per_xcd) / tiles);
      if (g < 8) g = 8;
      while (g > 8 && (g > smax || g > sbytes)) g -= 8;
      if (g <= smax && g <= sbytes) s = g;
    }
    if (s > smax) s = smax;
    if (s > sbytes) s = sbytes;
    if (s < 1) s = 1;
    const long r = vr_cdiv(vr_cdiv(M, s), 16) * 16;
    *rows = (int)r; *S = (int)vr_cdiv(M, r);
    return;
  }
  static const int wg_target = vr_tune("VRNET_WGRAD_TARGET", 1024);      // tuning aid
  auto splits = [&](long tiles) {
    long s = vr_cdiv(wg_target, tiles);
    // >= 512 rows per split on the big maps (the reduce pass is serial in S); down to 128 rows, at most 64
    // splits, on the small ones (M <= 8192), which otherwise cannot fill the chip
    long smax = vr_cdiv(M, 512);
    const long small = vr_cdiv(M, 128) < 64 ? vr_cdiv(M, 128) : 64;
    if (small > smax) smax = small;
    if (s > smax) s = smax;
    const long sbytes = (48L << 20) / (wsz * 4);
    if (s > sbytes) s = sbytes;
    return s < 1 ? 1 : s;
  };
  const int bn128 = Cin > 64 ? 128 : (Cin > 32 ? 64 : 32);
  const long tiles128 = vr_cdiv(Cout, 128) * vr_cdiv(Cin, bn128) * T;
  const long s128 = splits(tiles128);
  const long tiles64 = vr_cdiv(Cout, 64) * vr_cdiv(Cin, 64) * T;
  static const int force = vr_tune("VRNET_WGRAD_CFG", -1);   // tuning aid
  const bool small_ok = Cin > 32 && Cout > 32;
  // Measured (tools/tune_igemm.py --wgrad): 64 x 64 tiles win except for the large weight matrices whose
  // 128-wide tiling already yields >= 64 tiles (2560x640, 3x3 512x512, ...).
  if (bf16 || (force < 0 && tiles128 < 64 && small_ok) || (force == 1 && small_ok)) {
    *cfg = small_ok ? 1 : 0; *bn = 64;          // (bf16 without small_ok is rejected by the caller)
    *n_tiles = (int)vr_cdiv(Cout, 64); *c_tiles = (int)vr_cdiv(Cin, 64);
    const long s = splits(tiles64);
    const long q = bf16 ? 64 : 32;              // contraction rows per step of the bf16 / DMA kernels
    const long r = vr_cdiv(vr_cdiv(M, s), q) * q;
    *rows = (int)r; *S = (int)vr_cdiv(M, r);
  } else {
    *cfg = 0; *bn = bn128;
    *n_tiles = (int)vr_cdiv(Cout, 128); *c_tiles = (int)vr_cdiv(Cin, bn128);
    const long r = vr_cdiv(vr_cdiv(M, s128), BK) * BK;
    *rows = (int)r; *S = (int)vr_cdiv(M, r);
  }
}

// Slab reduction shared by every weight-gradient kernel: dw / dbias (+ layer-scale partials) from S slabs per stream.
// (also called from pgemm.hip: the plane weight gradient writes the same slabs)
int vr_wgrad_reduce_launch(float* slab, float* bslab, float* ls_part, long ls_stride, int S, int T, int Cout, int Cin,
                           int streams, const float* row_scale, float* dw, float* dbias, int accumulate,
                           const float* row_scale2, float* dw2, float* dbias2, const float* w, const float* w2,
                           const float* bias, const float* bias2, float* dls, float* dls2, hipStream_t st);
static int wgrad_reduce_launch(float* slab, float* bslab, float* ls_part, long ls_stride, int S, int T, int Cout, int Cin,
                               int streams, const float* row_scale, float* dw, float* dbias, int accumulate,
                               const float* row_scale2, float* dw2, float* dbias2, const float* w, const float* w2,
                               const float* bias, const float* bias2, float* dls, float* dls2, hipStream_t st) {
  return vr_wgrad_reduce_launch(slab, bslab, ls_part, ls_stride, S, T, Cout, Cin, streams, row_scale, dw, dbias, accumulate,
                                row_scale2, dw2, dbias2, w, w2, bias, bias2, dls, dls2, st);
}
int vr_wgrad_reduce_launch(float* slab, float* bslab, float* ls_part, long ls_stride, int S, int T, int Cout, int Cin,
                               int streams, const float* row_scale, float* dw, float* dbias, int accumulate,
                               const float* row_scale2, float* dw2, float* dbias2, const float* w, const float* w2,
                               const float* bias, const float* bias2, float* dls, float* dls2, hipStream_t st) {
  const bool rvec = (Cin % 4 == 0);                 // slabs are 16-byte aligned (workspace arena), rows of Cin floats
  if (dls && T == 1 && rvec && vr_aligned16(dw) && (!dw2 || vr_aligned16(dw2)) && vr_aligned16(w) && (!w2 || vr_aligned16(w2)) &&
      (!bias || dbias)) {
    // slab sums, row scale AND the layer-scale gradient in one launch: one workgroup per output row
    const int Q = Cin / 4;
    int sl = 1;
    while (sl < 16 && sl * 2 * Q <= 256) sl *= 2;
#define VR_WROWS(SL_)                                                                                                     \
  hipLaunchKernelGGL((wgrad_reduce_rows_kernel<SL_>), dim3(Cout, streams), dim3(256), 0, st, slab, bslab, row_scale, dw,  \
                     dbias, S, Cout, Cin, accumulate, row_scale2, dw2, dbias2, w, w2, bias, bias2, dls, dls2)
    switch (sl) {
      case 16: VR_WROWS(16); break;
      case 8: VR_WROWS(8); break;
      case 4: VR_WROWS(4); break;
      case 2: VR_WROWS(2); break;
      default: VR_WROWS(1); break;
    }
#undef VR_WROWS
    VR_LAUNCH_CHECK("conv2d_wgrad_reduce_rows");
    return VR_OK;
  }
  const long total = (long)T * Cout * Cin / (rvec ? 4 : 1) + (dbias ? Cout : 0);
#define VR_WREDUCE(VEC_, SL_)                                                                                        \
  hipLaunchKernelGGL((wgrad_reduce_kernel<VEC_, SL_>), dim3(vr_cdiv(total, 256 / SL_), streams), dim3(256), 0, st, slab, \
                     bslab, row_scale, dw, dbias, S, T, Cout, Cin, accumulate, row_scale2, dw2, dbias2, w, w2, ls_part)
  // lanes per output: enough of them to cover the serial slab loop of the small matrices, one thread per output
  // once the matrix alone yields >= 64K threads (a 16-lane block there is 6 K workgroups of 256 B of output each)
  const int sl = (total >= 65536 || S <= 2) ? 1 : ((total >= 16384 || S <= 8) ? 4 : 16);
  if (rvec) {
    if (sl == 16) VR_WREDUCE(4, 16);
    else if (sl == 4) VR_WREDUCE(4, 4);
    else VR_WREDUCE(4, 1);
  } else {
    if (sl == 16) VR_WREDUCE(1, 16);
    else if (sl == 4) VR_WREDUCE(1, 4);
    else VR_WREDUCE(1, 1);
  }
#undef VR_WREDUCE
  VR_LAUNCH_CHECK("conv2d_wgrad_reduce");
  if (dls) {
    hipLaunchKernelGGL(wgrad_rowdot_kernel, dim3(Cout, streams), dim3(64), 0, st, ls_part, bias, dls, Cout,
                       (int)((ls_stride - Cout) / Cout), ls_stride, accumulate, bias2, dls2);
    VR_LAUNCH_CHECK("conv2d_wgrad_rowdot");
  }
  return VR_OK;
}

// pair = 1: two-stream launch (the B samples are two streams of B/2, each with its own weight gradient)
extern "C" long vrnet_conv2d_wgrad_workspace(int B, int OH, int OW, int Cin, int Cout, int kh, int kw, int pair) {
  int cfg, bn, nt, ct, S, rows;
  const int streams = pair ? 2 : 1;
  const long Ms = (long)B * OH * OW / streams;
  wgrad_plan(Ms, Cin, Cout, kh * kw, &cfg, &bn, &nt, &ct, &S, &rows);
  long need = streams * ((long)S * kh * kw * Cout * Cin + (long)S * Cout) * 4 + 256;
  wgrad_plan(Ms, Cin, Cout, kh * kw, &cfg, &bn, &nt, &ct, &S, &rows, 1);      // the bf16 plan may split more
  const long need16 = streams * ((long)S * kh * kw * Cout * Cin + (long)S * Cout) * 4 + 256;
  if (need16 > need) need = need16;
  wgrad_plan(Ms, Cin, Cout, kh * kw, &cfg, &bn, &nt, &ct, &S, &rows, 2);      // so may the x6 plan
  if (cfg) {
    const long need6 = streams * ((long)S * kh * kw * Cout * Cin + (long)S * Cout) * 4 + 256;
    if (need6 > need) need = need6;
  }
  need += streams * ((long)Cout * Cin + Cout) * 4;      // layer-scale dot partials (1x1 convs, conv2d_wgrad dls)
  if (Cin <= 8 && Cout <= 8) {
    const long t = vr_tiny_wgrad_workspace((long)B * OH * OW, Cin, Cout, kh * kw);
    if (t > need) need = t;
  }
  if (!pair && vr_narrow_wgrad_ok(Cin, Cout, kh, kw, 1, 0)) {
    const long t = vr_narrow_wgrad_workspace((long)B * OH * OW, Cin, Cout);
    if (t > need) need = t;
  }
  return need;
}

extern "C" int vrnet_conv2d_wgrad_f32(const float* x, long ldx, const float* dy, long lddy, float* dw, float* dbias,
                                      const float* row_scale, int B, int H, int W, int Cin, int OH, int OW,
                                      int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                      int precision, float* dw2, float* dbias2, const float* row_scale2,
                                      const float* w, const float* bias, float* dls, const float* w2,
                                      const float* bias2, float* dls2, void* workspace, long workspace_bytes,
                                      void* stream) {
  VR_CHECK_ARG(x && dy && dw && workspace, "conv2d_wgrad: null tensor");
  if (vr_ablated("wgrad")) return VR_OK;
  {
    const long rows_ = (long)B * OH * OW;
    if (rows_ >= 32768 ? vr_ablated("wgrad_big") : (rows_ > 2048 ? vr_ablated("wgrad_mid") : vr_ablated("wgrad_small"))) return VR_OK;
  }
  const long M = (long)B * OH * OW;
  VR_CHECK_ARG(M < (1L << 31) && (long)B * H * W < (1L << 31), "conv2d_wgrad: too many pixels");
  const int streams = dw2 ? 2 : 1;
  VR_CHECK_ARG(streams == 1 || (B % 2 == 0 && (!dbias == !dbias2) && (!row_scale == !row_scale2)),
               "conv2d_wgrad: a two-stream launch needs an even batch and the second set of outputs");
  const int T = kh * kw;
  VR_CHECK_ARG(!dls || (w && T == 1 && (!bias || dbias) && (streams == 1 || (w2 && dls2 && (!bias == !bias2)))),
               "conv2d_wgrad: the layer-scale gradient needs a 1x1 conv, its weights, (with a bias) the bias gradient, "
               "and in a two-stream launch the second set");
  int cfg, bn, nt, ct, S, rows;
  const bool vec_all = (Cin % 4 == 0) && (ldx % 4 == 0) && vr_aligned16(x) && (Cout % 4 == 0) && (lddy % 4 == 0) &&
                       vr_aligned16(dy);
  int x6cfg = 0;
  if (precision == 2 || precision == 1) {      // the tile kernels: x6, or (precision 1) bf16-rounded operands
    if (vec_all) wgrad_plan(M / streams, Cin, Cout, T, &x6cfg, &bn, &nt, &ct, &S, &rows, 2);
    if (!x6cfg && precision == 2) precision = 0;       // no tile kernel for this shape: the fp32 MFMA path
  }
  if (!x6cfg) wgrad_plan(M / streams, Cin, Cout, T, &cfg, &bn, &nt, &ct, &S, &rows, precision == 1);
  const long need = vrnet_conv2d_wgrad_workspace(B, OH, OW, Cin, Cout, kh, kw, streams == 2);
  if (workspace_bytes < need) {
    vr_set_error("conv2d_wgrad: workspace %ld < %ld bytes", workspace_bytes, need);
    return VR_ERR_WORKSPACE;
  }
  VR_CHECK_ARG(streams == 1 || !tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride),
               "conv2d_wgrad: two-stream launch of a tiny-channel layer");
  if (tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride)) vr_note_kernel(4);
  if (tiny_shape(H, W, Cin, OH, OW, Cout, kh, kw, stride))
    return vr_tiny_wgrad(x, ldx, dy, lddy, dw, dbias, row_scale, B, H, W, Cin, Cout, kh, pad, dil, accumulate, workspace,
                         vr_stream(stream));
  if (vr_tune("VRNET_NARROW_WGRAD", 1) && streams == 1 && !dls && vr_narrow_wgrad_ok(Cin, Cout, kh, kw, stride, pad) && ldx % 4 == 0 && vr_aligned16(x)) {
    float *nslab, *nbslab;
    int nS;
    const int rc = vr_narrow_wgrad(x, ldx, dy, lddy, M, Cin, Cout, workspace, dbias != nullptr, &nslab, &nbslab, &nS,
                                   vr_stream(stream));
    if (rc) return rc;
    vr_note_kernel(5);
    return wgrad_reduce_launch(nslab, nbslab, nullptr, 0, nS, 1, Cout, Cin, 1, row_scale, dw, dbias, accumulate, nullptr, nullptr,
                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, vr_stream(stream));
  }
  WgradArgs p{};
  p.x = x; p.ldx = ldx; p.dy = dy; p.lddy = lddy;
  p.slab = reinterpret_cast<float*>(workspace);
  p.bslab = dbias ? p.slab + (long)streams * S * T * Cout * Cin : nullptr;
  p.M_half = (int)(M / streams);
  const bool rvec_ls = (Cin % 4 == 0);
  const long ls_stride = (long)T * Cout * Cin / (rvec_ls ? 4 : 1) + Cout;
  float* ls_part = dls ? p.slab + (long)streams * S * ((long)T * Cout * Cin + Cout) : nullptr;
  p.M = (int)M; p.OH = OH; p.OW = OW; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.dil = dil;
  p.rows_per_split = rows; p.n_tiles = nt; p.c_tiles = ct; p.splits = S;
  const bool vec = (Cin % 4 == 0) && (ldx % 4 == 0) && vr_aligned16(x) && (Cout % 4 == 0) && (lddy % 4 == 0) &&
                   vr_aligned16(dy);
  const bool ident = kh == 1 && kw == 1 && stride == 1 && pad == 0;
  hipStream_t st = vr_stream(stream);
  dim3 grid(nt * ct * T, S, streams), block(256);
#define VR_WGRAD(BM_, BN_, TM_, TN_, WM_, WN_)                                                                      \
  do {                                                                                                              \
    if (ident && vec) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, TM_, TN_, WM_, WN_, true, true>), grid, block, 0, st, p);   \
    else if (vec) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, TM_, TN_, WM_, WN_, false, true>), grid, block, 0, st, p);     \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, TM_, TN_, WM_, WN_, false, false>), grid, block, 0, st, p);             \
  } while (0)
  VR_CHECK_ARG(precision >= 0 && precision <= 2, "conv2d_wgrad: precision 0 (fp32 MFMA), 1 (bf16 operands) or 2 (x6)");
  if (x6cfg) {
    p.xcd_group = (S % 8 == 0 && (long)nt * ct * T * (S / 8) <= 96) ? 1 : 0;      // see wgrad_plan / wgrad_rows_xcd
    const dim3 grid8 = grid;
#define VR_WX6(TN_, TC_)                                                                                            \
  do {                                                                                                              \
    if (precision == 2) {                                                                                           \
      if (ident) hipLaunchKernelGGL((wgrad_x6_kernel<TN_, TC_, true, 6>), grid8, block, 0, st, p);                   \
      else hipLaunchKernelGGL((wgrad_x6_kernel<TN_, TC_, false, 6>), grid8, block, 0, st, p);                        \
    } else {                                                                                                        \
      if (ident) hipLaunchKernelGGL((wgrad_x6_kernel<TN_, TC_, true, 1>), grid8, block, 0, st, p);                   \
      else hipLaunchKernelGGL((wgrad_x6_kernel<TN_, TC_, false, 1>), grid8, block, 0, st, p);                        \
    }                                                                                                               \
  } while (0)
    if (x6cfg == 22) VR_WX6(2, 2);
    else if (x6cfg == 21) VR_WX6(2, 1);
    else VR_WX6(1, 2);
#undef VR_WX6
  } else if (precision == 1) {
    VR_CHECK_ARG(vec && cfg == 1 && rows % 64 == 0, "conv2d_wgrad: the bf16 path needs 16-byte aligned rows, channel counts that "
                                                    "are multiples of 4 and more than 32 channels on both sides");
    vr_wgrad_bf16_launch(&p, ident ? 1 : 0, nt * ct * T, S, streams, st);
  } else {
  static const int use_dma = vr_tune("VRNET_WGRAD_DMA", 1);   // tuning aid
  // measured (bench.py --detail): the ring wins only for the smallest weight matrices (<= 4 tiles: +5..19 %); with
  // more tiles the row-split grid already fills the chip and the 8-workgroups-per-CU kernel is 5-15 % faster
  if (cfg == 1 && vec && (use_dma == 2 || (use_dma && nt * ct * T <= 4))) {
    if (ident) hipLaunchKernelGGL((wgrad_dma_kernel<true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((wgrad_dma_kernel<false>), grid, block, 0, st, p);
  } else if (cfg == 1) VR_WGRAD(64, 64, 1, 1, 2, 2);
  else if (bn == 128) VR_WGRAD(128, 128, 2, 2, 2, 2);
  else if (bn == 64) VR_WGRAD(128, 64, 2, 1, 2, 2);
  else VR_WGRAD(128, 32, 1, 1, 4, 1);
  }
#undef VR_WGRAD
  vr_note_kernel(x6cfg ? (precision == 2 ? 6 : 3) : (precision == 1 ? 3 : 1));
  VR_LAUNCH_CHECK("conv2d_wgrad");
  return wgrad_reduce_launch(p.slab, p.bslab, ls_part, ls_stride, S, T, Cout, Cin, streams, row_scale, dw, dbias, accumulate,
                             row_scale2, dw2, dbias2, w, w2, bias, bias2, dls, dls2, st);
}

/* Size in bytes of the bf16 planes of a [J columns] x [K contraction] weight (K % 16 == 0). */
extern "C" long vrnet_conv_planes_bytes(int J, int K) { return (long)(K / 16) * (((J + 127) >> 7) << 1) * 6144; }

/* One launch that splits every weight of `table` (device array, 8 longs per entry: source address, J, K, element stride
 * between columns, element stride along the contraction, address of a per-contraction-index scale or 0, destination
 * address, index of the entry's first block; an entry has (K / 16) * 2 * ceil(J / 128) blocks) into three bf16 planes in the
 * stage-image order of igemm_planes_kernel.  total_blocks = sum of the entries' blocks. */
extern "C" int vrnet_conv_planes_pack_f32(const long* table, int nentries, long total_blocks, void* stream) {
  VR_CHECK_ARG(table && nentries > 0 && total_blocks > 0 && total_blocks < (1L << 31), "conv_planes_pack: bad table");
  hipLaunchKernelGGL(planes_pack_kernel, dim3((unsigned)total_blocks), dim3(128), 0, vr_stream(stream), table, nentries);
  VR_LAUNCH_CHECK("conv_planes_pack");
  return VR_OK;
}

extern "C" int vrnet_pack_weight_f32(const float* w_oihw, float* w_tnc, int Cout, int Cin, int kh, int kw,
                                     void* stream) {
  VR_CHECK_ARG(w_oihw && w_tnc, "pack_weight: null tensor");
  const long total = (long)kh * kw * Cout * Cin;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, vr_stream(stream), w_oihw, w_tnc,
                     Cout, Cin, kh * kw);
  VR_LAUNCH_CHECK("pack_weight");
  return VR_OK;
}
