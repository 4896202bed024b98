// Can the fragment splits of the x6 scheme (VALU) hide behind the bf16 MFMAs of the SAME wave?  Per K16 step of a 2x2
// register block: 24 MFMAs + 4 fragment splits (176 VALU operations).  Variants:
//   0 clustered   : splits of this step, then its MFMAs (what the compiler emits for the straightforward loop)
//   1 pipelined   : the splits of step i+1 in source order between the MFMAs of step i (sched_group_barrier 1 MFMA : 8 VALU)
//   2 / 3         : the same two with the accumulators pinned to AGPRs (inline asm, "+a")
//   hipcc --offload-arch=gfx950 -O3 x6_overlap.hip -o x6_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, bf16x8 (&out)[3]) {
  float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  unsigned p0[8], p1[8], p2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned b0 = __builtin_bit_cast(unsigned, x[e]);
    const float r1 = x[e] - __builtin_bit_cast(float, b0 & 0xffff0000u);
    const unsigned b1 = __builtin_bit_cast(unsigned, r1);
    const float r2 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
    p0[e] = b0; p1[e] = b1; p2[e] = __builtin_bit_cast(unsigned, r2);
  }
  u32x4 q0, q1, q2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    q0[e] = __builtin_amdgcn_perm(p0[2 * e + 1], p0[2 * e], 0x07060302u);
    q1[e] = __builtin_amdgcn_perm(p1[2 * e + 1], p1[2 * e], 0x07060302u);
    q2[e] = __builtin_amdgcn_perm(p2[2 * e + 1], p2[2 * e], 0x07060302u);
  }
  out[0] = __builtin_bit_cast(bf16x8, q0);
  out[1] = __builtin_bit_cast(bf16x8, q1);
  out[2] = __builtin_bit_cast(bf16x8, q2);
}

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& c, const bf16x8 a, const bf16x8 b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool AGPR>
__device__ __forceinline__ void x6(f32x16& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  mfma<AGPR>(c, a[1], b[1]); mfma<AGPR>(c, a[0], b[2]); mfma<AGPR>(c, a[2], b[0]);
  mfma<AGPR>(c, a[0], b[1]); mfma<AGPR>(c, a[1], b[0]); mfma<AGPR>(c, a[0], b[0]);
}

template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  constexpr bool AGPR = VAR >= 2, PIPE = (VAR & 1) != 0;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 fa[2][2], fb[2][2];
  for (int i = 0; i < 2; ++i) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) {
    fa[i][h][e] = seed + threadIdx.x * 1e-3f + i + h + e;
    fb[i][h][e] = seed - threadIdx.x * 1e-3f + i + h + e;
  }
  bf16x8 a3[2][3], b3[2][3], an[2][3], bn[2][3];
  for (int i = 0; i < 2; ++i) { split3(fa[i][0], fa[i][1], a3[i]); split3(fb[i][0], fb[i][1], b3[i]); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(fa[i][0]), "+v"(fa[i][1]), "+v"(fb[i][0]), "+v"(fb[i][1]));
    if (!PIPE) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { split3(fa[i][0], fa[i][1], a3[i]); split3(fb[i][0], fb[i][1], b3[i]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) x6<AGPR>(acc[i][j], a3[i], b3[j]);
    } else {
      // MFMAs of this step on (a3, b3); the next step's planes (an, bn) are produced meanwhile
#pragma unroll
      for (int i = 0; i < 2; ++i) { split3(fa[i][0], fa[i][1], an[i]); split3(fb[i][0], fb[i][1], bn[i]); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) x6<AGPR>(acc[i][j], a3[i], b3[j]);
      if (!AGPR) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { a3[i][p] = an[i][p]; b3[i][p] = bn[i][p]; }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
void run(int blocks) {
  float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double flop = (double)blocks * 4 * iters * 4 * 32768.0;
  printf("variant %d (%s, acc in %s) %d waves/SIMD: %7.3f ms %6.1f TF fp32-equivalent, %5.0f cyc@2.4GHz per K16 step per wave-slot\n", VAR,
         (VAR & 1) ? "pipelined" : "clustered", VAR >= 2 ? "AGPR" : "compiler's choice", blocks / 256, ms, flop / ms / 1e9,
         ms * 1e-3 * 2.4e9 / iters / (blocks / 256.0));
  (void)hipFree(out);
}
int main() {
  for (int b = 256; b <= 768; b += 256) { run<0>(b); run<1>(b); run<2>(b); run<3>(b); }
  return 0;
}
