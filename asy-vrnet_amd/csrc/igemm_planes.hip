// x6 GEMM for 1x1 convs with pre-split weights, activation fragments straight from global memory (round 6).
//
// igemm_planes_kernel (igemm.hip) brings BOTH operands into LDS by DMA and splits the A fragment after the barrier of every
// K16 step: ds_read -> ~50 VALU operations -> 12 MFMAs, one barrier per 384 matrix-pipe cycles, 8 KB of fp32 activations
// + 6 KB of weight planes through LDS per step.  In its 4 x 1 wave layout the 32 rows a wave multiplies are private to that
// wave, so the A operand needs no LDS at all: here lane (r, h) loads the 64 contiguous bytes k = 32 s + 16 h .. + 15 of its
// row with four global_load_dwordx4 (both lane halves of a row consume one 128-byte line per K32 step), one step ahead, into
// a landing buffer; the fragment splits run on the VALU in the shadow of the MFMAs of the step before, because they no
// longer wait behind a barrier.  The LDS ring holds the weight planes only (12 KB x TN per K32 stage), one barrier per 768
// matrix-pipe cycles, and 8 waves (256 rows) can share one B stage.
//
// Contraction order: MFMA k-index 8 h + i of sub-step j of step s is k = 32 s + 16 h + 8 j + i -- lane half h reads its B
// fragments from k16 step 2 s + h of the pack, half slot j; the pack (planes_pack_kernel) is unchanged.
#include "igemm_common.h"
#include "x6.h"

namespace {

// A-operand loads are inline asm: beside LDS-DMA hipcc waits vmcnt(0) for every ordinary VGPR-destination load, which would
// drain the ring each step.  The compiler does not count these loads and does not know that their destination is in flight:
// the ONLY variable they ever target is the landing buffer `land`, which no compiler-generated instruction touches between
// the load and the wait that names it (vr_wait_a: the destinations are "+v" operands of the wait, so every use sits behind
// it); the landed data is then copied to ordinary variables.
__device__ __forceinline__ void vr_load_a(f32x4 (&a)[4], const float* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[0]) : "v"(ptr) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(a[1]) : "v"(ptr) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(a[2]) : "v"(ptr) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(a[3]) : "v"(ptr) : "memory");
}
template <int N>
struct IC { static constexpr int value = N; };

template <int N>
__device__ __forceinline__ void vr_wait_a(f32x4 (&a)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}
// One statement for a wait whose count depends on (wave-uniform) run-time state: sel = 0: vmcnt(0), 1: vmcnt(N1), 2: vmcnt(N2).
// (Two vr_wait_a in the arms of an if / else are two definitions of `a` that meet in a PHI: the compiler then copies the
// registers -- in one arm BEFORE the s_waitcnt, i.e. in flight.  Seen in the ISA.)
template <int N1, int N2>
__device__ __forceinline__ void vr_wait_a_sel(f32x4 (&a)[4], const int sel) {
  asm volatile(
      "s_cmp_lg_u32 %4, 0\n\t"
      "s_cbranch_scc1 .LVRW%=_1\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_branch .LVRW%=_3\n"
      ".LVRW%=_1:\n\t"
      "s_cmp_lg_u32 %4, 1\n\t"
      "s_cbranch_scc1 .LVRW%=_2\n\t"
      "s_waitcnt vmcnt(%5)\n\t"
      "s_branch .LVRW%=_3\n"
      ".LVRW%=_2:\n\t"
      "s_waitcnt vmcnt(%6)\n"
      ".LVRW%=_3:"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
      : "s"(sel), "n"(N1), "n"(N2)
      : "memory", "scc");
}
// work = land, AFTER the wait that named `land`: the moves are asm too.  (A plain `work[i] = land[i]` behind vr_wait_a lets the
// register coalescer give the wait's tied operand the registers of `work` and copy `land` into them BEFORE the s_waitcnt --
// seen in the ISA of the first version: eight v_mov_b64 of the in-flight registers in front of the wait.)
__device__ __forceinline__ void vr_copy_landed(f32x4 (&work)[4], const f32x4 (&land)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a, b, c, d;
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"(land[i][0]), "v"(land[i][1]), "v"(land[i][2]), "v"(land[i][3]));
    work[i] = f32x4{a, b, c, d};
  }
}
template <int N>
__device__ __forceinline__ void vr_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one dword (two consecutive k) of each of the three planes
__device__ __forceinline__ void vr_split_pair(const float x0, const float x1, unsigned& q0, unsigned& q1, unsigned& q2) {
  const unsigned b00 = __builtin_bit_cast(unsigned, x0), b01 = __builtin_bit_cast(unsigned, x1);
  const float r10 = x0 - __builtin_bit_cast(float, b00 & 0xffff0000u), r11 = x1 - __builtin_bit_cast(float, b01 & 0xffff0000u);
  const unsigned b10 = __builtin_bit_cast(unsigned, r10), b11 = __builtin_bit_cast(unsigned, r11);
  const float r20 = r10 - __builtin_bit_cast(float, b10 & 0xffff0000u), r21 = r11 - __builtin_bit_cast(float, b11 & 0xffff0000u);
  q0 = __builtin_amdgcn_perm(b01, b00, 0x07060302u);
  q1 = __builtin_amdgcn_perm(b11, b10, 0x07060302u);
  q2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, r21), __builtin_bit_cast(unsigned, r20), 0x07060302u);
}

struct A3 { vr_u32x4 q[3]; };
__device__ __forceinline__ void vr_a3_planes(const A3& s, vr_bf16x8 (&out)[3]) {
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(vr_bf16x8, s.q[pl]);
}
// pairs [P0, P1) of the 8 values (lo, hi)
template <int P0, int P1>
__device__ __forceinline__ void vr_split_part(const f32x4 lo, const f32x4 hi, A3& s) {
#pragma unroll
  for (int e = P0; e < P1; ++e) {
    const float x0 = e < 2 ? lo[2 * e] : hi[2 * e - 4], x1 = e < 2 ? lo[2 * e + 1] : hi[2 * e - 3];
    unsigned q0, q1, q2;
    vr_split_pair(x0, x1, q0, q1, q2);
    // position pin: side-effect-free VALU code is emitted next to its users (for the planes of the NEXT step: the end of the loop
    // body, behind every MFMA); an empty volatile asm that takes the results keeps it in the MFMA group it was written beside
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
    s.q[0][e] = q0; s.q[1][e] = q1; s.q[2][e] = q2;
  }
}
// chunk c of CH equal chunks of a split (the chunk that rides with MFMA group c)
template <int CH>
__device__ __forceinline__ void vr_split_chunk(const int c, const f32x4 lo, const f32x4 hi, A3& s) {
  static_assert(CH == 2 || CH == 4, "two or four MFMA groups per sub-step");
  if constexpr (CH == 2) {
    if (c == 0) vr_split_part<0, 2>(lo, hi, s);
    else vr_split_part<2, 4>(lo, hi, s);
  } else {
    if (c == 0) vr_split_part<0, 1>(lo, hi, s);
    else if (c == 1) vr_split_part<1, 2>(lo, hi, s);
    else if (c == 2) vr_split_part<2, 3>(lo, hi, s);
    else vr_split_part<3, 4>(lo, hi, s);
  }
}

// TN: 64-column blocks per tile; NW: waves (32 rows each); WPS: waves per SIMD the register allocation must allow
// (workgroups per CU x NW / 4).  Ring of three B stages.
//
// Iteration s (K32 step s; `more` = step s + 2 exists), G = 4 TN MFMA groups of six:
//   top   wait: B(s) has landed (counted: what iteration s - 1 issued may be in flight) -> barrier
//   sub 0 groups 0 .. 2 TN - 1: MFMAs on planes s0 (split in iteration s - 1) and B fragments of k16 half 0; beside them the
//         split of work[2..3] (second half of A(s)), the B fragments of half 1, and behind each group its pieces of B(s + 2)
//   mid   wait: A(s + 1) has landed in `land` -> work = land (landed data: ordinary copies) -> load A(s + 2) into `land`
//   sub 1 groups: MFMAs on planes s1; beside them the split of work[0..1] (first half of A(s + 1)) = s0 of the next iteration
// The A load is issued in EVERY iteration (past the end it re-reads the last chunk: the landing buffer then has one def and
// one use per iteration, no PHI the compiler could copy in flight); the B pieces only while `more`.
template <int TN, int NW, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void igemm_planes_reg_kernel(const IgemmArgs p, const unsigned char* planes, int JB, int MT,
                                                                        int NT) {
  constexpr int NST = 3;
  constexpr int TNW = 2 * TN, BM = 32 * NW, BN = 64 * TN;
  constexpr int K16B = TN * 6144, ST_BYTES = 2 * K16B, NP = ST_BYTES / 1024;
  constexpr int PW = NP / NW, REM = NP % NW, PMAX = PW + (REM ? 1 : 0);      // wave w issues PW + (w < REM) B pieces per stage
  constexpr int G = 2 * TNW;                                                  // MFMA groups (of six) per K32 step
  static_assert(NST * ST_BYTES >= NW * 32 * STAGE_LD * 4, "epilogue staging must fit the ring");
  static_assert(PMAX <= G - 1, "one DMA piece behind each MFMA group but the first");
  // pieces issued in the slots of sub-step 0 (before the A load of the iteration): by a wave with PW / with PW + 1 pieces
  constexpr int C_LO = []() { int c = 0; for (int q = 0; q < PW; ++q) c += (1 + (q * (G - 1)) / PMAX) < TNW; return c; }();
  constexpr int C_HI = []() { int c = 0; for (int q = 0; q < PMAX; ++q) c += (1 + (q * (G - 1)) / PMAX) < TNW; return c; }();
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * ST_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int GT = 8 * ((MT + 7) >> 3) * NT;
  const int split = __builtin_amdgcn_readfirstlane(p.ksplit > 1 ? (int)blockIdx.x / GT : 0);
  const int L = blockIdx.x - split * GT, jj = L >> 3;
  const int nt = jj % NT, mt = (jj / NT) * 8 + (L & 7);
  if (mt >= MT) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nsteps_all = p.CK / 32;
  const int s_begin = p.ksplit > 1 ? (int)((long)split * nsteps_all / p.ksplit) : 0;
  const int nsteps = (p.ksplit > 1 ? (int)((long)(split + 1) * nsteps_all / p.ksplit) : nsteps_all) - s_begin;
  const int h = lane >> 5;
  const bool hi_wave = REM && wave < REM;      // this wave issues PW + 1 pieces per stage

  // ---- A: lane (r, h) of wave w owns k = 32 s + 16 h .. + 15 of row m0 + 32 w + r
  // (rows past M read row M - 1: an output row depends on its own A row only, and the epilogue stores rows < M)
  const float* a_ptr;
  {
    const int m = m0 + wave * 32 + (lane & 31);
    a_ptr = p.a + (long)(m < p.M ? m : p.M - 1) * p.lda + (long)s_begin * 32 + 16 * h;
  }
  int a_left = nsteps - 1;      // advances the pointer still has to make
  f32x4 land[4], work[4];
  auto load_a = [&]() __attribute__((always_inline)) {
    vr_load_a(land, a_ptr);
    a_ptr += a_left > 0 ? 32 : 0;
    --a_left;
  };

  // ---- B: piece q of a stage = 1 KB (q / (NP / 2): k16 step of the pair, q % (NP / 2): KB within its TN x 6 KB)
  const long b_step = 2L * JB * 6144;
  const unsigned char* b_src = planes + ((long)(n0 >> 6)) * 6144 + (long)lane * 16 + (long)s_begin * b_step;
  int ld_buf = 0;
  auto issue_piece = [&](int j) __attribute__((always_inline)) {      // this wave's j-th piece of the stage being issued
    if (j < PW || (REM && j == PW && wave < REM)) {
      const int q = wave + NW * j;
      const int kk = q / (NP / 2), within = q - kk * (NP / 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src + (long)kk * JB * 6144 + within * 1024),
                                       (__attribute__((address_space(3))) void*)(smem + ld_buf * ST_BYTES + q * 1024), 16, 0, 0);
    }
  };
  auto issue_end = [&]() __attribute__((always_inline)) {
    b_src += b_step;
    if (++ld_buf == NST) ld_buf = 0;
  };

  f32x16 acc[1][TNW];
#pragma unroll
  for (int j = 0; j < TNW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
  int b_off[TNW];      // + h * K16B is folded in: lane half h reads k16 step 2 s + h
#pragma unroll
  for (int i = 0; i < TNW; ++i) {
    const int rb = 32 * i + (lane & 31);
    b_off[i] = h * K16B + (rb >> 6) * 6144 + (rb & 63) * 16;       // + plane * 2048 + sub-step * 1024
  }

  // ---- prologue, in the issue order of an iteration: B(0) | A(0) | first pieces of B(1) | A(1) | rest of B(1)
  const bool two = nsteps > 1;
#pragma unroll
  for (int j = 0; j < PMAX; ++j) issue_piece(j);
  issue_end();
  load_a();
  if (two) {
#pragma unroll
    for (int q = 0; q < PMAX; ++q)
      if (1 + (q * (G - 1)) / PMAX < TNW) issue_piece(q);
  }
  // A(0) has landed: behind it came the first pieces of B(1)
  vr_wait_a_sel<C_LO, C_HI>(land, __builtin_amdgcn_readfirstlane(!two ? 0 : (hi_wave ? 2 : 1)));
  vr_copy_landed(work, land);
  load_a();
  if (two) {
#pragma unroll
    for (int q = 0; q < PMAX; ++q)
      if (1 + (q * (G - 1)) / PMAX >= TNW) issue_piece(q);
    issue_end();
  }
  A3 s0;      // planes of sub-step 0 of the step about to run
  vr_split_part<0, 4>(work[0], work[1], s0);

  int cur = 0;
  bool more_prev = two;      // iteration s - 1 issued a B stage
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 2 < nsteps;
    // B(s) has landed: what iteration s - 1 issued -- one A load and, if any, its B stage -- may still be in flight
    if (!more_prev) vr_wait_vm<4>();
    else if (hi_wave) vr_wait_vm<PW + 1 + 4>();
    else vr_wait_vm<PW + 4>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* Bs = smem + cur * ST_BYTES;
    // B fragments of sub-step 0 now; those of sub-step 1 column block by column block behind the MFMA group that frees the
    // registers of the same block (12 x TNW registers less than reading both up front)
    vr_bf16x8 bf[2][TNW][3];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[0][i][pl] = *reinterpret_cast<const vr_bf16x8*>(Bs + b_off[i] + pl * 2048);
    A3 s1, sn;
    vr_bf16x8 a3[3];
    vr_a3_planes(s0, a3);
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn) {
      acc[0][jn] = vr_mfma_x6(a3, bf[0][jn], acc[0][jn]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[1][jn][pl] = *reinterpret_cast<const vr_bf16x8*>(Bs + b_off[jn] + pl * 2048 + 1024);
      vr_split_chunk<TNW>(jn, work[2], work[3], s1);
      __builtin_amdgcn_sched_barrier(0);
      if (more && jn > 0) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q)
          if (1 + (q * (G - 1)) / PMAX == jn) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    vr_a3_planes(s1, a3);
    // A(s + 1) has landed: behind it came the rest of B(s + 1) (iteration s - 1) and this iteration's first pieces of B(s + 2) --
    // one whole stage of pieces in all; the last two iterations, which issue no pieces, simply drain
    vr_wait_a_sel<PW, PW + 1>(land, __builtin_amdgcn_readfirstlane(!more ? 0 : (hi_wave ? 2 : 1)));
    vr_copy_landed(work, land);
    load_a();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn) {
      acc[0][jn] = vr_mfma_x6(a3, bf[1][jn], acc[0][jn]);
      vr_split_chunk<TNW>(jn, work[0], work[1], sn);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q)
          if (1 + (q * (G - 1)) / PMAX == TNW + jn) issue_piece(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) issue_end();
    s0 = sn;
    more_prev = more;
    if (++cur == NST) cur = 0;
  }
  // the two A loads past the end target registers: they must have landed before the epilogue reuses them
  vr_wait_a<0>(land);
  __syncthreads();
  if constexpr (NW == 4) {
    if (p.ksplit > 1) {
      igemm_splitk_store<1, TNW>(p, acc, GT, L, split);
      return;
    }
  }
  igemm_epilogue<1, TNW, NW, 1>(p, acc, reinterpret_cast<float*>(smem), m0, n0);
}


// ------------------------------------------------------------------------------------------------
// Short contractions over the big maps (K = 64 / 128 at the 128 x 128 and 64 x 64 stages: 131 072 / 32 768 rows).  These
// launches move 50-170 MB for 1-4 GFLOP -- HBM-bound -- and ran at 1.5-2.7 TB/s on the tile kernels: a workgroup lives for two
// to eight K steps, so its prologue (first loads at HBM latency), its barriers and its epilogue are most of its life.  Here the
// whole B column tile (K / 16 x TN x 6 KB of weight planes) is brought into LDS ONCE per workgroup and stays; after that
// single barrier the four waves never synchronise again: each wave streams its own 32-row blocks of A through two landing
// buffers (two chunks = 8 KB per wave in flight, across block boundaries and under the epilogue), splits in the shadow of
// the MFMAs and stores its 32 x 64 TN outputs through a wave-private staging tile.
// Workgroup w: xcd = w % 8, slot = w / 8, column tile = slot % NT, row group = slot / NT -- the NT column tiles of a row
// group run on one XCD (one L2 fetches the A rows once); wave (row group q, wave) owns row blocks q * 4 + wave + k * stride.
template <int TN, int KS, int WPS>
__global__ __launch_bounds__(256, WPS) void igemm_planes_stream_kernel(const IgemmArgs p, const unsigned char* planes, int JB, int NT,
                                                                       int nrb, int RG) {
  constexpr int TNW = 2 * TN;
  constexpr int K16B = TN * 6144, ST_BYTES = 2 * K16B, B_BYTES = KS * ST_BYTES, NPIECE = B_BYTES / 1024;
  static_assert(KS == 2 || KS == 4, "the landing buffers alternate by step parity across blocks; chunk steps are named statically");
  __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 4 * 32 * STAGE_LD * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
  const int ct = slot % NT, rg = slot / NT;
  const int n0 = ct * 64 * TN;
  const int stride = 32 * RG;                      // row blocks between two consecutive blocks of a wave
  const int rb0 = (rg * 8 + xcd) * 4 + wave;
  const int nblk = rb0 < nrb ? (nrb - 1 - rb0) / stride + 1 : 0;
  const int h = lane >> 5;

  // ---- B: the whole column tile, once
  {
    const unsigned char* src = planes + ((long)(n0 >> 6)) * 6144 + (long)lane * 16;
#pragma unroll
    for (int q = wave; q < NPIECE; q += 4) {
      const int ks = q / (K16B / 1024), within = q - ks * (K16B / 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)ks * JB * 6144 + within * 1024),
                                       (__attribute__((address_space(3))) void*)(smem + q * 1024), 16, 0, 0);
    }
  }
  // ---- A: chunk (block i, step s) = 64 bytes per lane at row (rb0 + i stride) 32 + r, k = 32 s + 16 h
  const float* a_ptr = p.a + ((long)(nblk ? rb0 : 0) * 32 + (lane & 31)) * p.lda + 16 * h;
  const long blk_adv = (long)stride * 32 * p.lda - (long)(KS - 1) * 32;
  int ld_rem = nblk * KS;      // chunks still to load, the next one included
  f32x4 land[2][4];
  // the chunk with in-block step CS; then on to the next chunk -- past the last one the pointer stays (the two surplus loads of
  // the pipeline re-read the last chunk; advancing would read past the end of the tensor).  CS is static at every call site
  // (the step loop is unrolled): scalar selects, no branches.
  auto load_chunk = [&](f32x4 (&dst)[4], auto cs) __attribute__((always_inline)) {
    vr_load_a(dst, a_ptr);
    const long inc = decltype(cs)::value < KS - 1 ? 32L : blk_adv;
    a_ptr += ld_rem > 1 ? inc : 0L;
    ld_rem -= ld_rem > 1 ? 1 : 0;
  };
  load_chunk(land[0], IC<0>());
  load_chunk(land[1], IC<1>());
  vr_wait_a<0>(land[0]);
  vr_wait_a<0>(land[1]);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (nblk == 0) return;

  int b_off[TNW];
#pragma unroll
  for (int i = 0; i < TNW; ++i) {
    const int rbn = 32 * i + (lane & 31);
    b_off[i] = h * K16B + (rbn >> 6) * 6144 + (rbn & 63) * 16;       // + step * ST_BYTES + plane * 2048 + sub-step * 1024
  }
  float* stage = reinterpret_cast<float*>(smem + B_BYTES) + wave * (32 * STAGE_LD);
  A3 s0, s1;
  vr_split_part<0, 4>(land[0][0], land[0][1], s0);
  vr_bf16x8 bf[TNW][3];
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bf[i][pl] = *reinterpret_cast<const vr_bf16x8*>(smem + b_off[i] + pl * 2048);

  for (int blk = 0; blk < nblk; ++blk) {
    f32x16 acc[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      f32x4(&cur)[4] = land[s & 1];
      f32x4(&nxt)[4] = land[(s & 1) ^ 1];
      const unsigned char* B1 = smem + s * ST_BYTES + 1024;                          // sub-step 1 of this step
      const unsigned char* B0n = smem + ((s + 1) % KS) * ST_BYTES;                   // sub-step 0 of the next step
      vr_bf16x8 a3[3], bfn[TNW][3];
      // sub-step 0: MFMAs on s0; beside them the split of the chunk's second half and the B fragments of sub-step 1
      vr_a3_planes(s0, a3);
#pragma unroll
      for (int jn = 0; jn < TNW; ++jn) {
        acc[jn] = vr_mfma_x6(a3, bf[jn], acc[jn]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bfn[jn][pl] = *reinterpret_cast<const vr_bf16x8*>(B1 + b_off[jn] + pl * 2048);
        vr_split_chunk<TNW>(jn, cur[2], cur[3], s1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the chunk in `cur` is consumed: its buffer takes chunk s + 2; then chunk s + 1 must have landed (only the load just
      // issued may be in flight)
      if ((s + 2) % KS == 0) load_chunk(cur, IC<0>());
      else if ((s + 2) % KS == 1) load_chunk(cur, IC<1>());
      else if ((s + 2) % KS == 2) load_chunk(cur, IC<2>());
      else load_chunk(cur, IC<3>());
      vr_wait_a<4>(nxt);
      vr_a3_planes(s1, a3);
#pragma unroll
      for (int jn = 0; jn < TNW; ++jn) {
        acc[jn] = vr_mfma_x6(a3, bfn[jn], acc[jn]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[jn][pl] = *reinterpret_cast<const vr_bf16x8*>(B0n + b_off[jn] + pl * 2048);
        vr_split_chunk<TNW>(jn, nxt[0], nxt[1], s0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int row0 = (rb0 + blk * stride) * 32;
    // (named one by one: a loop around a call that contains wave barriers is not unrolled, and acc[j] would go to scratch)
    igemm_epilogue_tile(p, acc[0], stage, row0, n0);
    igemm_epilogue_tile(p, acc[1], stage, row0, n0 + 32);
    if constexpr (TNW == 4) {
      igemm_epilogue_tile(p, acc[2], stage, row0, n0 + 64);
      igemm_epilogue_tile(p, acc[3], stage, row0, n0 + 96);
    }
  }
  // the surplus loads target registers: drain before the wave ends
  vr_wait_a<0>(land[0]);
  vr_wait_a<0>(land[1]);
}

}  // namespace

// internal entry used by vrnet_conv2d_f32 (igemm.hip): variant = 100 * NW + 10 * TN + workgroups per CU.  Returns 0 when
// launched, 1 when the shape has no such kernel (the caller falls back to igemm_planes_kernel).
int vr_igemm_planes_reg_launch(const void* args, const void* planes, int variant, long M, int S, hipStream_t st) {
  const IgemmArgs& p = *reinterpret_cast<const IgemmArgs*>(args);
  if (p.CK % 32 != 0) return 1;
  const int JB = (int)(((p.CN + 127) >> 7) << 1);
  const unsigned char* pl = reinterpret_cast<const unsigned char*>(planes);
#define VR_PR(TN_, NW_, WPS_)                                                                                         \
  do {                                                                                                                \
    const long mt = vr_cdiv(M, 32 * NW_), ntile = vr_cdiv(p.CN, 64 * TN_);                                            \
    dim3 grid((unsigned)(8 * vr_cdiv(mt, 8) * ntile * S));                                                            \
    hipLaunchKernelGGL((igemm_planes_reg_kernel<TN_, NW_, WPS_>), grid, dim3(64 * NW_), 0, st, p, pl, JB, (int)mt,   \
                       (int)ntile);                                                                                   \
    return 0;                                                                                                         \
  } while (0)
  switch (variant) {
    case 413: VR_PR(1, 4, 3);
    case 414: VR_PR(1, 4, 4);
    case 422: VR_PR(2, 4, 2);
    case 423: VR_PR(2, 4, 3);
    case 812: if (S == 1) VR_PR(1, 8, 4); else return 1;
    case 821: if (S == 1) VR_PR(2, 8, 2); else return 1;
    default: return 1;
  }
#undef VR_PR
}

// Resident-B streaming kernel for K = 64 / 128 over >= 32 768 rows (variant 0: the caller's rule).  Returns 0 when launched.
int vr_igemm_planes_stream_launch(const void* args, const void* planes, long M, hipStream_t st) {
  const IgemmArgs& p = *reinterpret_cast<const IgemmArgs*>(args);
  if ((p.CK != 64 && p.CK != 128) || M % 32 != 0 || !p.e_vec || p.ksplit > 1 || p.CN % 4 != 0) return 1;
  const int JB = (int)(((p.CN + 127) >> 7) << 1);
  const unsigned char* pl = reinterpret_cast<const unsigned char*>(planes);
  const int nrb = (int)(M / 32);
  // column tiles of 64 (the 128-wide form -- 64 accumulator registers more -- spills at two waves per SIMD)
  const int TN = 1;
  const int NT = (int)vr_cdiv(p.CN, 64 * TN);
  const int per_cu = 2;      // 234 registers: two waves per SIMD (the LDS -- 43 / 67 KB -- would admit three / two)
  // waves per column tile: as many as fit (workgroups in multiples of 8 per column tile), each with the same number of blocks
  long wg_ct = (256L * per_cu / NT) / 8 * 8;
  if (wg_ct < 8) wg_ct = 8;
  const long k = vr_cdiv(nrb, 4 * wg_ct);
  const int RG = (int)vr_cdiv(vr_cdiv(nrb, k), 32);      // row groups per XCD
  dim3 grid((unsigned)(8L * RG * NT));
  if (p.CK == 64) hipLaunchKernelGGL((igemm_planes_stream_kernel<1, 2, 2>), grid, dim3(256), 0, st, p, pl, JB, NT, nrb, RG);
  else hipLaunchKernelGGL((igemm_planes_stream_kernel<1, 4, 2>), grid, dim3(256), 0, st, p, pl, JB, NT, nrb, RG);
  return 0;
}
