// Plane GEMMs (round 4): the x6 product scheme with BOTH operands already split into bf16 planes in HBM.
//
// Rounds 2-3 split the fp32 operands inside the GEMM kernels (x6.h: vr_split3, ~44 VALU operations per 8-element
// fragment); on gfx950 an MFMA holds the SIMD's vector issue for 8 of its 32 cycles and everything else a wave issues in
// an MFMA gap has to fit the other 24, so the splits, the address arithmetic of the fp32 images and the LDS-DMA issue all
// came out of matrix-pipe time (0.20-0.25 of the x6 ceiling in the step).  Here a tensor that is only ever a
// GEMM operand -- the GroupNorm outputs, the Cluster output, the Mlp hidden activation, the gradients [df | dv], du and
// the block output gradients -- is WRITTEN as three bf16 planes by its producer (t = p0 + p1 + p2 exactly, vr_store_planes4
// in igemm_common.h: the split runs once per element in an HBM-bound kernel instead of once per element, column tile and
// consumer in the GEMMs), the weights are split once per step as before, and the GEMM main loop is DMA + ds_read_b128 +
// MFMA only: no VALU work on operands at all.
//
//   plane tensor "p3": element (r, k) of plane q at base[q * plane + r * ld + k], bf16; ld % 8 == 0, 16-byte aligned.
//   np = 3: the fp32 value, six products a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1 (x6.h); np = 1: a bf16 tensor, one product
//   (compute_dtype "bf16": the same kernels on bf16 activations).
//
// pgemm_kernel: C[M][N] = A[M][K] . B[N][K]^T + fused epilogue (igemm_epilogue_tile), 128 x 128 x 32 tile, 512 threads =
// 8 waves as 2 (rows) x 2 (columns) x 2 (halves of the 32-deep stage): a wave owns a 64 x 64 block of the tile (2 x 2
// accumulators) and one K16 step of every stage, i.e. 12 fragment reads (ds_read_b128) and 24 MFMAs per stage with NP = 3;
// the two waves of a SIMD (w, w + 4) work on the same 64 x 64 block and meet once at the end through LDS.  Three stages of
// 48 KB (one workgroup per CU): operands go global -> LDS by global_load_lds_dwordx4 (six 1 KB pieces per wave and stage:
// waves 0-3 the A planes, waves 4-7 the B planes), XOR swizzle on the source address as in igemm_dma_kernel (64-byte rows:
// slot (r, c ^ ((r >> 2) & 3)) -> conflict-free ds_read_b128).  Pipeline: at the top of iteration s the fragments of stage s
// are in registers and stage s + 1 has landed; the wave issues the fragment reads of stage s + 1 into its second register
// set, then its 24 MFMAs with the DMA pieces of stage s + 3 between them (into the slot of stage s, which nobody reads any
// more), waits for stage s + 2 (counted vmcnt: stage s + 3 stays in flight) and meets the others at ONE raw s_barrier.
// LDS latency, DMA latency and the barrier skew all sit behind the MFMAs of the SIMD's other wave.
#include "igemm_common.h"
#include "x6.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned int vp_zero_page[64];      // 256 zero bytes: source of masked rows

struct PlaneOps {
  const unsigned short* a; long lda; long a_plane;      // A planes: rows m (GEMM rows), contraction k contiguous
  const unsigned short* b; long ldb; long b_plane;      // B planes: rows n (GEMM columns), contraction k contiguous
};

template <int NP>
struct PFrag {
  vr_bf16x8 a[2][NP];
  vr_bf16x8 b[2][NP];
};

template <int NP>
__device__ __forceinline__ f32x16 vp_products(const vr_bf16x8 (&a)[NP], const vr_bf16x8 (&b)[NP], f32x16 c) {
  if constexpr (NP == 3) {
    return vr_mfma_x6(a, b, c);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  }
}

constexpr int PG_BM = 128, PG_BN = 128, PG_BK = 32, PG_NST = 3;
constexpr int PG_PLANE = 128 * 64;                      // one plane image of a stage: 128 rows x 64 bytes

// DBG (diagnostic build only): bit 0 = no MFMAs, bit 1 = no DMA inside the loop (the ring keeps the prologue's stages),
// bit 2 = no fragment reads inside the loop: timing ablations, results garbage.
template <int NP, int DBG = 0>
__global__ __launch_bounds__(512, NP == 1 ? 4 : 2) void pgemm_kernel(const IgemmArgs p, const PlaneOps o, int MT, int NT) {
  constexpr int OPER = NP * PG_PLANE;                  // one operand of a stage
  constexpr int ST_BYTES = 2 * OPER;                   // 48 KB (NP = 3) / 16 KB (NP = 1)
  constexpr int P = 2 * NP;                            // DMA pieces per wave and stage
  constexpr int XCH = 4 * 2 * 32 * 256;                // K-half exchange: 4 wave pairs x 2 directions x 32 registers x 256 B
  // np = 3: the ring (144 KB) covers exchange + staging.  np = 1: the ring is 48 KB; the staging tiles REUSE the exchange area
  // (one more barrier) so that a workgroup holds 64 KB and two fit a CU: with bf16 operands a tile is a few microseconds of
  // matrix work, and a second workgroup is what covers its prologue and its store-heavy epilogue.
  constexpr int STG = 8 * 32 * STAGE_LD * 4;
  constexpr int LDS_BYTES = NP == 1 ? (XCH > PG_NST * ST_BYTES ? XCH : PG_NST * ST_BYTES)
                                    : ((PG_NST * ST_BYTES > XCH + STG) ? PG_NST * ST_BYTES : XCH + STG);
  static_assert(STG <= XCH, "staging tiles must fit the exchange area");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int L = blockIdx.x, jj = L >> 3;
  const int nt = jj % NT, mt = (jj / NT) * 8 + (L & 7);      // the NT column tiles of a row tile: consecutive ids on ONE XCD
  if (mt >= MT) return;
  const int m0 = mt * PG_BM, n0 = nt * PG_BN;
  const int nsteps = p.CK / PG_BK;

  // ---- DMA roles: waves 0-3 fill the A planes, waves 4-7 the B planes; wave w (mod 4) owns rows 32 w .. 32 w + 31 of its
  // operand's 128 rows: two 16-row pieces x NP planes.  Lane l of a piece lands at byte 16 l of the piece: row l >> 2, slot
  // l & 3, which must hold chunk c = slot ^ ((row >> 2) & 3) of that row (the swizzle the fragment reads undo).
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(vp_zero_page);
  asm volatile("" : "+s"(zero_page));
  const bool isA = wave < 4;
  const unsigned char* src[2];
  int inc[2];
  long pl_bytes;
  {
    const unsigned short* base = isA ? o.a : o.b;
    const long ld = isA ? o.lda : o.ldb;
    pl_bytes = 2 * (isA ? o.a_plane : o.b_plane);
    const int rmax = isA ? p.M : p.CN, r0 = isA ? m0 : n0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int r = (wave & 3) * 32 + g * 16 + (lane >> 2);       // row in the tile
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      const bool ok = r0 + r < rmax;
      src[g] = ok ? reinterpret_cast<const unsigned char*>(base + (long)(r0 + r) * ld) + 16 * c : zero_page;
      inc[g] = ok ? 2 * PG_BK : 0;
    }
  }
  const bool ok0 = inc[0] != 0, ok1 = inc[1] != 0;
  int ld_buf = 0;
  // piece i of the stage being issued: group g = i & 1, plane i >> 1
  auto issue_piece = [&](int i) {
    const int g = i & 1, q = i >> 1;
    const bool ok = g ? ok1 : ok0;
    const unsigned char* s = src[g] + (ok ? q * pl_bytes : 0);
    unsigned char* d = smem + ld_buf * ST_BYTES + (isA ? 0 : OPER) + q * PG_PLANE + ((wave & 3) * 2 + g) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d,
                                     16, 0, 0);
  };
  auto issue_end = [&]() {
    src[0] += inc[0];
    src[1] += inc[1];
    if (++ld_buf == PG_NST) ld_buf = 0;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses: lane (row lane & 31, half h) reads the 8 contraction values 16 wk + 8 h .. + 7 = chunk 2 wk + h
  const int h = lane >> 5;
  const int ra = wm * 64 + (lane & 31), rb = wn * 64 + (lane & 31);
  const int a_off = ra * 64 + ((((2 * wk + h) ^ ((ra >> 2) & 3))) << 4);               // + 2048 i + PG_PLANE q
  const int b_off = OPER + rb * 64 + ((((2 * wk + h) ^ ((rb >> 2) & 3))) << 4);
  auto read_frags = [&](PFrag<NP>& f, int buf) {
    const unsigned char* st = smem + buf * ST_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        f.a[i][q] = *reinterpret_cast<const vr_bf16x8*>(st + a_off + 2048 * i + PG_PLANE * q);
        f.b[i][q] = *reinterpret_cast<const vr_bf16x8*>(st + b_off + 2048 * i + PG_PLANE * q);
      }
  };

  // ---- prologue: up to three stages in flight; stages 0 and 1 must have landed before the loop
#pragma unroll
  for (int st = 0; st < PG_NST; ++st)
    if (st < nsteps) {
#pragma unroll
      for (int i = 0; i < P; ++i) issue_piece(i);
      issue_end();
    }
  if (nsteps >= 3) __builtin_amdgcn_s_waitcnt(NP == 3 ? 0x0F76 : 0x0F72);      // vmcnt(P): stages 0 and 1 have landed
  else __builtin_amdgcn_s_waitcnt(0x0F70);                                      // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  PFrag<NP> F0, F1;
  int rd_buf = 0;
  read_frags(F0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();      // every wave holds its stage-0 fragments: slot 0 may be refilled
  __builtin_amdgcn_sched_barrier(0);

  // One iteration: Fc = fragments of stage s (in registers), Fn receives those of stage s + 1.  The waits are the
  // compiler's own s_waitcnt (builtin, not inline asm): its scoreboard must know that Fn has arrived when the loop comes
  // round, or it parks an lgkmcnt(0) in front of the first MFMA of the next iteration -- behind the reads just issued.
  //   simm16 = vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
  constexpr int W_LGKM0 = 0xC07F;                                  // lgkmcnt(0)
  constexpr int W_LGKM0_VM_P = NP == 3 ? 0x0076 : 0x0072;          // lgkmcnt(0) + vmcnt(P): one stage stays in flight
  constexpr int W_LGKM0_VM0 = 0x0070;                              // lgkmcnt(0) + vmcnt(0)
  auto mfmas = [&](PFrag<NP>& Fc, bool more) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = t >> 1, j = t & 1;
      if constexpr (DBG & 1) {
#pragma unroll
        for (int q = 0; q < NP; ++q) asm volatile("" ::"v"(Fc.a[i][q]), "v"(Fc.b[j][q]));
      } else {
        acc[i][j] = vp_products<NP>(Fc.a[i], Fc.b[j], acc[i][j]);
      }
      if (more && !(DBG & 2)) {      // the DMA pieces of the stage three ahead, spread behind the MFMA groups
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q % 4 == t) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more && !(DBG & 2)) issue_end();
  };
  // steady state: a stage three ahead exists (so do s + 1 and s + 2)
  auto step_main = [&](PFrag<NP>& Fc, PFrag<NP>& Fn) {
    if (++rd_buf == PG_NST) rd_buf = 0;
    if constexpr (!(DBG & 4)) read_frags(Fn, rd_buf);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Fc, true);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(W_LGKM0_VM_P);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // the last (up to) three iterations and short contractions
  auto step_tail = [&](PFrag<NP>& Fc, PFrag<NP>& Fn, int s) {
    const bool has_next = s + 1 < nsteps;
    const bool more = s + PG_NST < nsteps;
    if (has_next) {
      if (++rd_buf == PG_NST) rd_buf = 0;
      read_frags(Fn, rd_buf);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Fc, more);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nsteps && !more) __builtin_amdgcn_s_waitcnt(W_LGKM0_VM0);      // stage s + 2 is the last one in flight
    else if (more) __builtin_amdgcn_s_waitcnt(W_LGKM0_VM_P);
    else __builtin_amdgcn_s_waitcnt(W_LGKM0);
    if (has_next) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  int s = 0;
  for (; s + 2 + PG_NST <= nsteps; s += 2) {      // both iterations of the pair have a stage three ahead
    step_main(F0, F1);
    step_main(F1, F0);
  }
  for (; s + 1 < nsteps; s += 2) {
    step_tail(F0, F1, s);
    step_tail(F1, F0, s + 1);
  }
  if (s < nsteps) step_tail(F0, F1, s);
  __syncthreads();      // every wave is done with the ring: the exchange and staging areas may overwrite it

  // ---- the two K halves of a 64 x 64 block meet: wave (wk = 0) finishes row block 0, its partner (wk = 1) row block 1; each
  // hands the other row block's two accumulators over through LDS (a + b = b + a: the order of the two halves cannot matter)
  {
    const int pair = wave & 3;
    float* mine = reinterpret_cast<float*>(smem) + (pair * 2 + wk) * (32 * 64);          // what this wave hands over
    const float* theirs = reinterpret_cast<const float*>(smem) + (pair * 2 + (wk ^ 1)) * (32 * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wk ? acc[0][j][r + e] : acc[1][j][r + e];
        *reinterpret_cast<f32x4*>(mine + ((j * 4 + (r >> 2)) * 64 + lane) * 4) = v;
      }
    __syncthreads();
    f32x16 fin[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + ((j * 4 + (r >> 2)) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[j][r + e] = (wk ? acc[1][j][r + e] : acc[0][j][r + e]) + v[e];
      }
    if constexpr (NP == 1) __syncthreads();      // every wave has read its partner's half: the exchange area becomes the staging tiles
    float* stage = reinterpret_cast<float*>(smem + (NP == 1 ? 0 : XCH)) + wave * (32 * STAGE_LD);
    const int row0 = m0 + wm * 64 + wk * 32, col0 = n0 + wn * 64;
    igemm_epilogue_tile(p, fin[0], stage, row0, col0);
    igemm_epilogue_tile(p, fin[1], stage, row0, col0 + 32);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient on plane operands: dW[n][c] = sum_m dy[m][n] x[m][c] for a 1x1 conv, dy and x both bf16 planes (np = 3: the
// six products; np = 1: bf16 tensors).  The in-kernel-split weight gradient (wgrad_x6_kernel) splits FOUR fragments per 24
// MFMAs -- both operands are activations -- and reads every fragment with 8 ds_read_b32: its VALU share exceeds what the MFMA
// gaps hide.  Here a stage is 32 contraction rows of the 128 (n) + 128 (c) columns, row-major as in HBM (256-byte rows per
// plane, 48 KB per stage, three stages), and the fragments -- 8 consecutive ROWS of one column -- come out of gfx950's
// transposing ds_read_b64_tr_b16 (two per fragment and plane): 24 reads, 24 MFMAs, no VALU per stage and wave.  Same 8-wave
// geometry as pgemm_kernel: 2 (n) x 2 (c) x 2 (halves of the 32 rows), the halves meet through LDS at the end.  LDS image of a
// plane: row r at 256 r, its 16-byte unit u at slot u ^ (4 (r & 3)): the four rows a transposing read touches per 16-lane
// group then sit in four different 64-byte quarters of the bank row (conflict-free); the swizzle is applied to the DMA's
// source address.  Output: the fp32 slabs of the other weight-gradient kernels (one per row split, reduced in a fixed order
// by wgrad_reduce_kernel); the bias gradient (column sums of dy) rides on the matrix pipe: dy^T times a fragment of ones.
struct PwgradArgs {
  const unsigned short* x; long ldx; long x_plane;
  const unsigned short* dy; long lddy; long dy_plane;
  float* slab; float* bslab;
  int M, Cin, Cout, rows_per_split, splits, n_tiles, c_tiles;
};

template <int NP>
__global__ __launch_bounds__(512, NP == 1 ? 4 : 2) void pwgrad_kernel(const PwgradArgs p) {
  constexpr int PLANE = 32 * 256;                      // one plane image of a stage: 32 rows x 256 bytes
  constexpr int OPER = NP * PLANE, ST_BYTES = 2 * OPER, P = 2 * NP;
  constexpr int XCH = 4 * 2 * 32 * 256;
  constexpr int LDS_BYTES = PG_NST * ST_BYTES > XCH + 2048 ? PG_NST * ST_BYTES : XCH + 2048;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  // all tiles of one row split on ONE XCD (ids b, b + 8, ... share an L2): the split's dy / x rows enter one L2 once
  const int tiles = p.n_tiles * p.c_tiles;
  const int L = blockIdx.x, j = L >> 3;
  const int split = (j / tiles) * 8 + (L & 7), tile = j % tiles;
  if (split >= p.splits) return;
  const int nt = tile / p.c_tiles, ct = tile - nt * p.c_tiles;
  const int n0 = nt * 128, c0 = ct * 128;
  const int m_begin = split * p.rows_per_split, m_end = min(p.M, m_begin + p.rows_per_split);
  const int nsteps = (m_end - m_begin + 31) >> 5;
  const bool do_bias = p.bslab != nullptr && ct == 0;

  // ---- DMA roles: waves 0-3 fill the dy planes, waves 4-7 the x planes; wave w (mod 4) owns rows 8 w .. 8 w + 7 of the
  // stage: two 4-row pieces x NP planes.  Lane l of a piece: row l >> 4, slot l & 15 <- unit slot ^ (4 (row & 3)).
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(vp_zero_page);
  asm volatile("" : "+s"(zero_page));
  const bool isY = wave < 4;
  const unsigned short* base = isY ? p.dy : p.x;
  const long ld = isY ? p.lddy : p.ldx;
  const long pl_bytes = 2 * (isY ? p.dy_plane : p.x_plane);
  const int cmax = isY ? p.Cout : p.Cin, col0 = isY ? n0 : c0;
  int row_in_stage[2];
  long col_byte;
  bool col_ok;
  {
    const int u = (lane & 15) ^ (4 * ((lane >> 4) & 3));      // (row & 3) == (lane >> 4) & 3: pieces start at multiples of 4 rows
    col_ok = col0 + 8 * u < cmax;                             // (channel counts are multiples of 8: a unit is in or out as a whole)
    col_byte = 2L * (col0 + 8 * u);
    row_in_stage[0] = (wave & 3) * 8 + (lane >> 4);
    row_in_stage[1] = row_in_stage[0] + 4;
  }
  int ld_m = m_begin, ld_buf = 0;
  auto issue_piece = [&](int i) {
    const int g = i & 1, q = i >> 1;
    const int m = ld_m + row_in_stage[g];
    const bool ok = col_ok && m < m_end;
    const unsigned char* s = ok ? reinterpret_cast<const unsigned char*>(base + (long)m * ld) + col_byte + q * pl_bytes : zero_page;
    unsigned char* d = smem + ld_buf * ST_BYTES + (isY ? 0 : OPER) + q * PLANE + ((wave & 3) * 2 + g) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)d,
                                     16, 0, 0);
  };
  auto issue_end = [&]() {
    ld_m += 32;
    if (++ld_buf == PG_NST) ld_buf = 0;
  };

  // bacc: the bias gradient of ONE 32-channel block per wave -- wave (wm, wn, wk) takes block wn of its 64 output channels
  // (both fragments are in its registers anyway) over its half wk of the rows
  f32x16 acc[2][2], bacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; bacc[r] = 0.f;
  }
  // ---- transposed fragment reads: lane = (half h, 16-lane group g, row q of the group's 4, column quad pp); it supplies the
  // address of row 16 wk + 8 h + q (+ 4 for the second read), columns cb + 16 g + 4 pp .. + 3 of a 32-column block at cb
  const int h = lane >> 5, g = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int rrow = 16 * wk + 8 * h + q4;                      // (rrow & 3) == q4, also for rrow + 4
  auto tr_off = [&](int cb) {                                 // byte offset inside a plane image
    const int unit = (cb >> 3) + 2 * g + (pp >> 1);
    return rrow * 256 + ((unit ^ (4 * q4)) << 4) + 8 * (pp & 1);
  };
  const int y_off0 = tr_off(wm * 64), y_off1 = tr_off(wm * 64 + 32);
  const int x_off0 = OPER + tr_off(wn * 64), x_off1 = OPER + tr_off(wn * 64 + 32);
  typedef __attribute__((address_space(3))) vr_bf16x4* lds4;
  auto frag = [&](const unsigned char* st, int off) {
    const vr_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(st + off));
    const vr_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(st + off + 4 * 256));
    const vr_bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return f;
  };
  auto read_frags = [&](PFrag<NP>& f, int buf) {
    const unsigned char* st = smem + buf * ST_BYTES;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f.a[0][q] = frag(st, y_off0 + PLANE * q);
      f.a[1][q] = frag(st, y_off1 + PLANE * q);
      f.b[0][q] = frag(st, x_off0 + PLANE * q);
      f.b[1][q] = frag(st, x_off1 + PLANE * q);
    }
  };
  const __bf16 one = (__bf16)1.0f;
  const vr_bf16x8 ones = {one, one, one, one, one, one, one, one};

  if constexpr (NP == 1) {
    // bf16 tensors: a stage is 8 MFMAs per SIMD -- two workgroups per CU (64 KB of LDS, <= 128 registers) cover each other's
    // waits, so the loop is the plain ring: wait for stage s, ONE barrier, refill the slot read last time, fragments, MFMAs
#pragma unroll
    for (int st = 0; st < PG_NST - 1; ++st)
      if (st < nsteps) {
#pragma unroll
        for (int i = 0; i < P; ++i) issue_piece(i);
        issue_end();
      }
    PFrag<NP> F;
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0F72);      // vmcnt(P): stage s + 1 may still be in flight
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool more = s + PG_NST - 1 < nsteps;
      read_frags(F, cur);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = t >> 1, jj = t & 1;
        acc[i][jj] = vp_products<NP>(F.a[i], F.b[jj], acc[i][jj]);
        if (more && t < P) issue_piece(t);
      }
      if (more) issue_end();
      if (do_bias) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wn ? F.a[1][0] : F.a[0][0], ones, bacc, 0, 0, 0);
      if (++cur == PG_NST) cur = 0;
    }
  } else {
#pragma unroll
  for (int st = 0; st < PG_NST; ++st)
    if (st < nsteps) {
#pragma unroll
      for (int i = 0; i < P; ++i) issue_piece(i);
      issue_end();
    }
  if (nsteps >= 3) __builtin_amdgcn_s_waitcnt(NP == 3 ? 0x0F76 : 0x0F72);
  else __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  PFrag<NP> F0, F1;
  int rd_buf = 0;
  if (nsteps > 0) read_frags(F0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  constexpr int W_LGKM0 = 0xC07F, W_LGKM0_VM_P = NP == 3 ? 0x0076 : 0x0072, W_LGKM0_VM0 = 0x0070;
  auto mfmas = [&](PFrag<NP>& Fc, bool more) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = t >> 1, jj = t & 1;
      acc[i][jj] = vp_products<NP>(Fc.a[i], Fc.b[jj], acc[i][jj]);
      if (more) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q % 4 == t) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) issue_end();
    if (do_bias) {      // column sums of dy: every plane times ones, small planes first (exact products)
#pragma unroll
      for (int q = NP - 1; q >= 0; --q)
        bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wn ? Fc.a[1][q] : Fc.a[0][q], ones, bacc, 0, 0, 0);
    }
  };
  auto step_main = [&](PFrag<NP>& Fc, PFrag<NP>& Fn) {
    if (++rd_buf == PG_NST) rd_buf = 0;
    read_frags(Fn, rd_buf);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Fc, true);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(W_LGKM0_VM_P);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step_tail = [&](PFrag<NP>& Fc, PFrag<NP>& Fn, int s) {
    const bool has_next = s + 1 < nsteps;
    const bool more = s + PG_NST < nsteps;
    if (has_next) {
      if (++rd_buf == PG_NST) rd_buf = 0;
      read_frags(Fn, rd_buf);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(Fc, more);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nsteps && !more) __builtin_amdgcn_s_waitcnt(W_LGKM0_VM0);
    else if (more) __builtin_amdgcn_s_waitcnt(W_LGKM0_VM_P);
    else __builtin_amdgcn_s_waitcnt(W_LGKM0);
    if (has_next) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  int s = 0;
  for (; s + 2 + PG_NST <= nsteps; s += 2) {
    step_main(F0, F1);
    step_main(F1, F0);
  }
  for (; s + 1 < nsteps; s += 2) {
    step_tail(F0, F1, s);
    step_tail(F1, F0, s + 1);
  }
  if (s < nsteps) step_tail(F0, F1, s);
  }
  __syncthreads();

  // ---- the two halves of the contraction meet (as in pgemm_kernel); wave (wk = 0) stores n-block 0, its partner n-block 1
  const int pair = wave & 3;
  float* mine = reinterpret_cast<float*>(smem) + (pair * 2 + wk) * (32 * 64);
  const float* theirs = reinterpret_cast<const float*>(smem) + (pair * 2 + (wk ^ 1)) * (32 * 64);
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = wk ? acc[0][jj][r + e] : acc[1][jj][r + e];
      *reinterpret_cast<f32x4*>(mine + ((jj * 4 + (r >> 2)) * 64 + lane) * 4) = v;
    }
  float* bx = reinterpret_cast<float*>(smem + XCH);      // bias partials of the wk = 1 waves: [pair][2 blocks][32 rows... via acc map]
  if (do_bias && wk == 1 && (lane & 31) == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bx[(wm * 2 + wn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = bacc[r];
  }
  __syncthreads();
  float* slab = p.slab + (long)split * p.Cout * p.Cin;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int c = c0 + wn * 64 + 32 * jj + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; r += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(theirs + ((jj * 4 + (r >> 2)) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sum = (wk ? acc[1][jj][r + e] : acc[0][jj][r + e]) + v[e];
        const int n = n0 + wm * 64 + 32 * wk + ((r + e) & 3) + 8 * ((r + e) >> 2) + 4 * h;
        if (n < p.Cout && c < p.Cin) slab[(long)n * p.Cin + c] = sum;
      }
    }
  }
  if (do_bias && wk == 0 && (lane & 31) == 0) {      // every column of the ones product holds the same sums
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int n = n0 + (wm * 2 + wn) * 32 + row;
      if (n < p.Cout) p.bslab[(long)split * p.Cout + n] = bacc[r] + bx[(wm * 2 + wn) * 32 + row];
    }
  }
}

// Split of fp32 matrices into bf16 planes.  Block = 32 rows x 64 k of one matrix; thread = (row, 8 consecutive k).
__device__ __forceinline__ void planes_split_block(const float* src, long R, long K, long sr, long sk, const float* ksc,
                                                   unsigned short* dst, long ld, long plane, long blk, int np) {
  const long kblocks = (K + 63) >> 6;
  const long rb = blk / kblocks, kb = blk - rb * kblocks;
  const long r = rb * 32 + (threadIdx.x >> 3), k0 = kb * 64 + (threadIdx.x & 7) * 8;
  if (r >= R || k0 >= K) return;
  float v[8];
  if (sk == 1 && k0 + 8 <= K && ((reinterpret_cast<uintptr_t>(src + r * sr + k0) & 15) == 0)) {
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src + r * sr + k0), hi4 = *reinterpret_cast<const f32x4*>(src + r * sr + k0 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = lo4[i]; v[4 + i] = hi4[i]; }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (k0 + i < K) ? src[r * sr + (k0 + i) * sk] : 0.f;
  }
  if (ksc) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (k0 + i < K) v[i] *= ksc[k0 + i];
  }
  unsigned short* d = dst + r * ld + k0;
  const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  vr_store_planes4(d, plane, np, a);
  // K % 8 == 4: the thread at k0 = K - 4 owns only four columns -- a second store would land in the next row's first four
  // (racing with that row's own writer) or, on the last row, past the plane
  if (k0 + 4 < K) vr_store_planes4(d + 4, plane, np, b);
}

// Multi-tensor form: one launch for every weight of a step.  Table entry e (10 longs): source address, rows R, contraction
// K, source element strides (row, k), address of a scale per k or 0, destination address, destination row stride,
// destination plane stride (elements), first block.
__global__ __launch_bounds__(256) void planes_split_kernel(const long* table, int nentries, int np) {
  int e = 0;
  {
    int lo = 0, hi = nentries - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[10 * mid + 9] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    e = lo;
  }
  const long* t = table + 10 * e;
  planes_split_block(reinterpret_cast<const float*>(t[0]), t[1], t[2], t[3], t[4], reinterpret_cast<const float*>(t[5]),
                     reinterpret_cast<unsigned short*>(t[6]), t[7], t[8], (long)blockIdx.x - t[9], np);
}

// One row-major matrix (an activation tensor whose producer has no plane output).
__global__ __launch_bounds__(256) void planes_split_one_kernel(const float* src, long R, long K, long sr, unsigned short* dst,
                                                               long ld, long plane, int np) {
  planes_split_block(src, R, K, sr, 1, nullptr, dst, ld, plane, (long)blockIdx.x, np);
}

}  // namespace

/* Whether vrnet_gemm_planes_f32 has a kernel for a product of `rows` x `cols` over a contraction of K (1 / 0).  The
 * kernel runs one 128 x 128 tile per workgroup and per CU. */
extern "C" int vrnet_gemm_planes_ok(long rows, int cols, int K) {
  static const int min_tiles = vr_tune("VRNET_PGEMM_MIN_TILES", 96);
  static const int min_cols = vr_tune("VRNET_PGEMM_MIN_COLS", 96);
  return (K % PG_BK == 0 && K >= PG_BK && cols >= min_cols && cols % 4 == 0 && vr_cdiv(rows, PG_BM) * vr_cdiv(cols, PG_BN) >= min_tiles) ? 1 : 0;
}

/* y[M][N] = epilogue(A[M][K] . B[N][K]^T): the GEMM of a 1x1 convolution (forward: A = activations, B = w[Cout][Cin]; data
 * gradient: A = dy, B = the transposed weights with the layer scale folded in) on operands that already ARE bf16 planes
 * (header of this file).  np = 3: fp32 values as three planes each, six products; np = 1: bf16 tensors.  The epilogue is
 * vrnet_conv2d_f32's: bias, aux (x gelu'), ypre, act (0 none, 1 ReLU, 2 GELU), res (+ res_scale), accumulate, stats (fp64
 * (sum, sumsq) pairs per 32 x 32 tile), colstats; the result goes to `y` (fp32, may be null) and / or to `yp` as planes
 * (yp_np = 3: exact split, the next GEMM's operand; 1: bf16).  half_side bit 0: `ypre` is a bf16 tensor (stored rounded), bit 1:
 * `aux` is a bf16 tensor (row strides in elements) -- the Mlp's pre-activation u in bf16 mode.  Replaces, for operands in plane form, the 1x1 launches of
 * vrnet_conv2d_f32 (backbone/fusion/vr_coc.py:145-147, 187, 205-207 and their autograd). */
extern "C" int vrnet_gemm_planes_f32(const void* a, long lda, long a_plane, const void* b, long ldb, long b_plane, int np,
                                     long M, int N, int K, const float* bias, float* y, long ldy, void* yp, long ldyp,
                                     long yp_plane, int yp_np, int act, float* ypre, long ldypre, const float* res, long ldres,
                                     const float* res_scale, const float* aux, long ldaux, int accumulate, double* stats,
                                     long stats_hw, const vrnet_conv_colstats* colstats, int half_side, void* stream) {
  VR_CHECK_ARG(a && b && (y || yp), "gemm_planes: null tensor");
  VR_CHECK_ARG(np == 1 || np == 3, "gemm_planes: np = 3 (fp32 as three bf16 planes) or 1 (bf16)");
  VR_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && K >= PG_BK && K % PG_BK == 0, "gemm_planes: bad shape (K %% 32 == 0)");
  VR_CHECK_ARG(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && vr_aligned16(a) && vr_aligned16(b) &&
                   (np == 1 || (a_plane % 8 == 0 && b_plane % 8 == 0)),
               "gemm_planes: operand rows must be 16-byte aligned (row and plane strides %% 8 == 0)");
  VR_CHECK_ARG(N % 4 == 0 && (!y || (ldy >= N && ldy % 4 == 0 && vr_aligned16(y))) && (!bias || vr_aligned16(bias)) &&
                   (!ypre || (ldypre % 4 == 0 && vr_aligned16(ypre))) && (!res || (ldres % 4 == 0 && vr_aligned16(res))) &&
                   (!res_scale || vr_aligned16(res_scale)) && (!aux || (ldaux % 4 == 0 && vr_aligned16(aux))),
               "gemm_planes: the epilogue needs N %% 4 == 0 and 16-byte aligned rows");
  VR_CHECK_ARG(!yp || ((yp_np == 1 || yp_np == 3) && ldyp >= N && ldyp % 4 == 0 && (reinterpret_cast<uintptr_t>(yp) & 7) == 0 &&
                       (yp_np == 1 || yp_plane % 4 == 0)),
               "gemm_planes: plane output needs 8-byte aligned rows");
  VR_CHECK_ARG(!accumulate || y, "gemm_planes: accumulate needs the fp32 output");
  VR_CHECK_ARG(!stats || (stats_hw > 0 && stats_hw % 32 == 0 && N > 32), "gemm_planes: output statistics need > 32 columns and "
                                                                          "samples of a multiple of 32 rows");
  if (vr_ablated("igemm")) return VR_OK;
  IgemmArgs p{};
  p.bias = bias; p.y = y; p.ldy = ldy; p.ypre = ypre; p.ldypre = ldypre; p.res = res; p.ldres = ldres; p.res_scale = res_scale;
  p.aux = aux; p.ldaux = ldaux; p.act = act; p.accumulate = accumulate; p.M = (int)M; p.CN = N; p.CK = K; p.e_vec = 1;
  p.ypre_bf16 = half_side & 1; p.aux_bf16 = (half_side >> 1) & 1;      // (8-byte rows suffice for those: checked as 16-byte above)
  p.MH = 1; p.MW = (int)M; p.stats = stats; p.stats_nb = (int)vr_cdiv(N, 32);
  p.yp = reinterpret_cast<unsigned short*>(yp); p.ldyp = ldyp; p.yp_plane = yp_plane; p.yp_np = yp_np;
  if (colstats) {
    VR_CHECK_ARG(colstats->partial && (!colstats->tile_totals || colstats->gamma) && N > 32 &&
                     (!colstats->x2 || (colstats->ldx2 % 4 == 0 && vr_aligned16(colstats->x2))),
                 "gemm_planes: bad column statistics");
    p.col_part = colstats->partial; p.col_x2 = colstats->x2; p.ld_col_x2 = colstats->ldx2;
    p.col_gamma = colstats->gamma; p.col_tot = colstats->tile_totals;
  }
  PlaneOps o{reinterpret_cast<const unsigned short*>(a), lda, a_plane, reinterpret_cast<const unsigned short*>(b), ldb, b_plane};
  const long mt = vr_cdiv(M, PG_BM), nt = vr_cdiv(N, PG_BN);
  dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * nt)), block(512);
  hipStream_t st = vr_stream(stream);
#ifdef VR_TUNING
  switch (np == 3 ? vr_tune("VRNET_PGEMM_DBG", 0) : 0) {
    case 1: hipLaunchKernelGGL((pgemm_kernel<3, 1>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 2: hipLaunchKernelGGL((pgemm_kernel<3, 2>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 3: hipLaunchKernelGGL((pgemm_kernel<3, 3>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 4: hipLaunchKernelGGL((pgemm_kernel<3, 4>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 5: hipLaunchKernelGGL((pgemm_kernel<3, 5>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 6: hipLaunchKernelGGL((pgemm_kernel<3, 6>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    case 7: hipLaunchKernelGGL((pgemm_kernel<3, 7>), grid, block, 0, st, p, o, (int)mt, (int)nt); return VR_OK;
    default: break;
  }
#endif
  if (np == 3) hipLaunchKernelGGL((pgemm_kernel<3>), grid, block, 0, st, p, o, (int)mt, (int)nt);
  else hipLaunchKernelGGL((pgemm_kernel<1>), grid, block, 0, st, p, o, (int)mt, (int)nt);
  vr_note_kernel(np == 3 ? 10 : 11);
  VR_LAUNCH_CHECK("gemm_planes");
  return VR_OK;
}

/* Blocks of entry (R rows, K contraction) in vrnet_planes_split_f32's grid. */
extern "C" long vrnet_planes_split_blocks(long R, long K) { return vr_cdiv(R, 32) * vr_cdiv(K, 64); }

/* fp32 matrices -> bf16 planes (np = 3: exact three-way split, round-to-nearest-even; np = 1: rounded to bf16), one launch
 * for a whole table of matrices: the weights of a step (forward: w[Cout][Cin] as it is; data gradient: the transpose, row
 * stride 1 / k stride Cin, with the layer scale as `scale per k`), or one activation tensor.  `table`: device array of
 * nentries x 10 longs (source address, R, K, source row stride, source k stride, scale address or 0, destination address,
 * destination row stride, destination plane stride, first block = running sum of vrnet_planes_split_blocks). */
extern "C" int vrnet_planes_split_f32(const long* table, int nentries, long total_blocks, int np, void* stream) {
  VR_CHECK_ARG(table && nentries > 0 && total_blocks > 0 && total_blocks < (1L << 31) && (np == 1 || np == 3), "planes_split: bad arguments");
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)total_blocks), dim3(256), 0, vr_stream(stream), table, nentries, np);
  VR_LAUNCH_CHECK("planes_split");
  return VR_OK;
}

/* One row-major fp32 matrix (R rows of K values, row stride lds) -> bf16 planes: the conversion pass for an activation
 * tensor whose producer has no plane output. */
extern "C" int vrnet_planes_from_f32(const float* src, long lds, long R, long K, void* dst, long ld, long plane, int np,
                                     void* stream) {
  VR_CHECK_ARG(src && dst && R > 0 && K > 0 && lds >= K && ld >= K && (np == 1 || np == 3), "planes_from_f32: bad arguments");
  VR_CHECK_ARG(ld % 4 == 0 && K % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 && (np == 1 || plane % 4 == 0),
               "planes_from_f32: destination rows must be 8-byte aligned");
  const long blocks = vr_cdiv(R, 32) * vr_cdiv(K, 64);
  VR_CHECK_ARG(blocks < (1L << 31), "planes_from_f32: too large");
  hipLaunchKernelGGL(planes_split_one_kernel, dim3((unsigned)blocks), dim3(256), 0, vr_stream(stream), src, R, K, lds,
                     reinterpret_cast<unsigned short*>(dst), ld, plane, np);
  VR_LAUNCH_CHECK("planes_from_f32");
  return VR_OK;
}

// ---- weight gradient on plane operands
int vr_wgrad_reduce_launch(float* slab, float* bslab, float* ls_part, long ls_stride, int S, int T, int Cout, int Cin,
                           int streams, const float* row_scale, float* dw, float* dbias, int accumulate,
                           const float* row_scale2, float* dw2, float* dbias2, const float* w, const float* w2,
                           const float* bias, const float* bias2, float* dls, float* dls2, hipStream_t st);

static void pwgrad_plan(long M, int Cin, int Cout, int np, int* nt, int* ct, int* S, int* rows) {
  *nt = (int)vr_cdiv(Cout, 128); *ct = (int)vr_cdiv(Cin, 128);
  const long tiles = (long)*nt * *ct;
  static const int wgs = vr_tune("VRNET_PWGRAD_WGS", 256);
  const int target = np == 1 ? 2 * wgs : wgs;                      // workgroups per CU: one (np = 3), two (np = 1)
  long s = 8 * (target / (8 * tiles));                             // multiples of 8: a split's tiles share an XCD
  if (s < 8) s = vr_cdiv(target, tiles);
  const long smax = vr_cdiv(M, 128), sbytes = (48L << 20) / ((long)Cout * Cin * 4);
  if (s > smax) s = smax;
  if (s > sbytes) s = sbytes;
  if (s < 1) s = 1;
  const long r = vr_cdiv(vr_cdiv(M, s), 32) * 32;
  *rows = (int)r; *S = (int)vr_cdiv(M, r);
}

/* Whether vrnet_wgrad_planes_f32 takes a weight gradient of M contraction rows, Cin x Cout channels (1 / 0). */
extern "C" int vrnet_wgrad_planes_ok(long M, int Cin, int Cout) {
  static const int min_c = vr_tune("VRNET_PWGRAD_MIN_C", 64);      // (64 channels fill half a tile: such layers have >= 32 768 rows and stream from HBM)
  return (M >= 256 && Cin >= min_c && Cout >= min_c && Cin % 8 == 0 && Cout % 8 == 0) ? 1 : 0;
}

extern "C" long vrnet_wgrad_planes_workspace(long M, int Cin, int Cout) {
  int nt, ct, S, rows;
  pwgrad_plan(M, Cin, Cout, 1, &nt, &ct, &S, &rows);      // (the np = 1 plan has the most splits)
  return ((long)S * ((long)Cout * Cin + Cout) + (long)Cout * Cin + Cout) * 4 + 256;
}

/* Weight (+ bias, + layer-scale) gradient of a 1x1 conv on plane operands: dw[Cout][Cin] (+)= row_scale[n] * sum_m dy[m][n] *
 * x[m][c]; dbias, row_scale, accumulate, (w, bias, dls) as vrnet_conv2d_wgrad_f32.  x: planes of the conv's input (M rows of Cin),
 * dy: planes of the output gradient (M rows of Cout), same np (3: fp32 values, six products; 1: bf16 tensors).  Channel counts
 * multiples of 8.  Replaces the in-kernel-split weight gradient for operands whose producers wrote planes (autograd of
 * backbone/fusion/vr_coc.py:145-147, 187, 205-207).  Kernel families 12 (np 3) / 13 (np 1). */
extern "C" int vrnet_wgrad_planes_f32(const void* x, long ldx, long x_plane, const void* dy, long lddy, long dy_plane, int np,
                                      long M, int Cin, int Cout, float* dw, float* dbias, const float* row_scale, int accumulate,
                                      const float* w, const float* bias, float* dls, void* workspace, long workspace_bytes,
                                      void* stream) {
  VR_CHECK_ARG(x && dy && dw && workspace && (np == 1 || np == 3), "wgrad_planes: bad arguments");
  VR_CHECK_ARG(M > 0 && M < (1L << 31) && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && ldx >= Cin && lddy >= Cout &&
                   ldx % 8 == 0 && lddy % 8 == 0 && vr_aligned16(x) && vr_aligned16(dy) && (np == 1 || (x_plane % 8 == 0 && dy_plane % 8 == 0)),
               "wgrad_planes: channel counts, row and plane strides must be multiples of 8 (16-byte rows)");
  VR_CHECK_ARG(!dls || (w && (!bias || dbias)), "wgrad_planes: the layer-scale gradient needs the weights and (with a bias) the bias gradient");
  if (vr_ablated("wgrad")) return VR_OK;
  int nt, ct, S, rows;
  pwgrad_plan(M, Cin, Cout, np, &nt, &ct, &S, &rows);
  const long need = vrnet_wgrad_planes_workspace(M, Cin, Cout);
  if (workspace_bytes < need) {
    vr_set_error("wgrad_planes: workspace %ld < %ld bytes", workspace_bytes, need);
    return VR_ERR_WORKSPACE;
  }
  PwgradArgs p{};
  p.x = reinterpret_cast<const unsigned short*>(x); p.ldx = ldx; p.x_plane = x_plane;
  p.dy = reinterpret_cast<const unsigned short*>(dy); p.lddy = lddy; p.dy_plane = dy_plane;
  p.slab = reinterpret_cast<float*>(workspace);
  p.bslab = dbias ? p.slab + (long)S * Cout * Cin : nullptr;
  p.M = (int)M; p.Cin = Cin; p.Cout = Cout; p.rows_per_split = rows; p.splits = S; p.n_tiles = nt; p.c_tiles = ct;
  const long ls_stride = (long)Cout * Cin / 4 + Cout;
  float* ls_part = dls ? p.slab + (long)S * ((long)Cout * Cin + Cout) : nullptr;
  hipStream_t st = vr_stream(stream);
  dim3 grid((unsigned)(8 * vr_cdiv(S, 8) * nt * ct)), block(512);
  if (np == 3) hipLaunchKernelGGL((pwgrad_kernel<3>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((pwgrad_kernel<1>), grid, block, 0, st, p);
  vr_note_kernel(np == 3 ? 12 : 13);
  VR_LAUNCH_CHECK("wgrad_planes");
  return vr_wgrad_reduce_launch(p.slab, p.bslab, ls_part, ls_stride, S, 1, Cout, Cin, 1, row_scale, dw, dbias, accumulate, nullptr,
                                nullptr, nullptr, w, nullptr, bias, nullptr, dls, nullptr, st);
}
