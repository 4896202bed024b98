// Practical ceiling of the "x6" inner loop on the box, register-resident: per K16 step of a 2x2 (or 2x1) register block,
// split the fp32 fragments into three bf16 planes (VALU) and issue the 6 bf16 MFMAs per 32x32 block.
//   what = 0: MFMAs only, 1: splits only, 2: both.   hipcc --offload-arch=gfx950 -O3 x6_peak.hip -o x6_peak.bin
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// packed-fp32 subtraction (v_pk_add_f32: two elements per instruction)
__device__ __forceinline__ void split3_pk(const f32x4 lo4, const f32x4 hi4, bf16x8 (&out)[3]) {
  f32x2 x[4] = {{lo4[0], lo4[1]}, {lo4[2], lo4[3]}, {hi4[0], hi4[1]}, {hi4[2], hi4[3]}};
  u32x4 q0, q1, q2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32x2 b0 = __builtin_bit_cast(u32x2, x[e]);
    const u32x2 m0 = {b0[0] & 0xffff0000u, b0[1] & 0xffff0000u};
    const f32x2 r1 = x[e] - __builtin_bit_cast(f32x2, m0);
    const u32x2 b1 = __builtin_bit_cast(u32x2, r1);
    const u32x2 m1 = {b1[0] & 0xffff0000u, b1[1] & 0xffff0000u};
    const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, m1);
    const u32x2 b2 = __builtin_bit_cast(u32x2, r2);
    q0[e] = __builtin_amdgcn_perm(b0[1], b0[0], 0x07060302u);
    q1[e] = __builtin_amdgcn_perm(b1[1], b1[0], 0x07060302u);
    q2[e] = __builtin_amdgcn_perm(b2[1], b2[0], 0x07060302u);
  }
  out[0] = __builtin_bit_cast(bf16x8, q0);
  out[1] = __builtin_bit_cast(bf16x8, q1);
  out[2] = __builtin_bit_cast(bf16x8, q2);
}
#ifdef PK
#define split3 split3_pk
#endif

template <int TM, int TN, int WHAT>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 fa[TM][2], fb[TN][2];
  for (int i = 0; i < TM; ++i) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) fa[i][h][e] = seed + threadIdx.x * 1e-3f + i + h + e;
  for (int i = 0; i < TN; ++i) for (int h = 0; h < 2; ++h) for (int e = 0; e < 4; ++e) fb[i][h][e] = seed - threadIdx.x * 1e-3f + i + h + e;
  bf16x8 a3[TM][3], b3[TN][3];
  for (int i = 0; i < TM; ++i) split3(fa[i][0], fa[i][1], a3[i]);
  for (int i = 0; i < TN; ++i) split3(fb[i][0], fb[i][1], b3[i]);
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (WHAT != 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        asm volatile("" : "+v"(fa[i][0]), "+v"(fa[i][1]));      // a fresh fragment every step as far as the compiler knows
        split3(fa[i][0], fa[i][1], a3[i]);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        asm volatile("" : "+v"(fb[i][0]), "+v"(fb[i][1]));
        split3(fb[i][0], fb[i][1], b3[i]);
      }
    }
    if (WHAT != 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][2], b3[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) asm volatile("" :: "v"(a3[i][p]));
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) asm volatile("" :: "v"(b3[i][p]));
    }
  }
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = sink;
}

template <int TM, int TN, int WHAT>
void run(int blocks) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<TM, TN, WHAT>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * TM * TN * 32768.0;       // fp32-equivalent flops (one product per six MFMAs)
    const double cyc = ms * 1e-3 * 2.4e9 / iters / (blocks / 256.0);          // cycles @2.4 GHz per step per wave-slot
    if (rep == 2)
      printf("%dx%d %-6s %d waves/SIMD: %7.3f ms  %6.1f TF fp32-equivalent, %6.0f cyc@2.4GHz per K16 step per wave\n", TM, TN,
             WHAT == 0 ? "mfma" : WHAT == 1 ? "split" : "both", blocks / 256, ms, flop / ms / 1e9, cyc);
  }
  hipFree(out);
}
int main() {
  for (int b = 256; b <= 768; b += 256) {
    if (b == 256) { run<2, 2, 0>(b); run<2, 2, 1>(b); run<2, 2, 2>(b); run<2, 1, 0>(b); run<2, 1, 1>(b); run<2, 1, 2>(b); }
    if (b == 512) { run<2, 2, 0>(b); run<2, 2, 1>(b); run<2, 2, 2>(b); run<2, 1, 0>(b); run<2, 1, 1>(b); run<2, 1, 2>(b); }
    if (b == 768) { run<2, 2, 0>(b); run<2, 2, 1>(b); run<2, 2, 2>(b); run<2, 1, 0>(b); run<2, 1, 1>(b); run<2, 1, 2>(b); }
  }
  return 0;
}
