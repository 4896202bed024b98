// fp32 products on the bf16 matrix pipe ("x6"): operand split, rounding and the six-product MFMA group shared by the
// implicit-GEMM kernels (igemm.hip) and the fused MLP kernels (mlp_fused.hip).
#pragma once
#include "common.h"

namespace {

// ---- fp32 products on the bf16 matrix pipe ("x6"): every fp32 operand is split EXACTLY into three bf16 values
// a = a0 + a1 + a2 (truncation: a0 = top 16 bits of a, a1 = top 16 bits of a - a0, a2 = a - a0 - a1, which has at most
// 8 significant bits left), and a*b is accumulated in fp32 as the six bf16 x bf16 products (exact in fp32) with
// i + j <= 2: a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1.  Every plane keeps 8 significant bits, so with truncation
// |a1| < 2^-7 |a| and |a2| < 2^-14 |a|: the dropped products a1b2 and a2b1 are each below 2^-21 |ab| (a2b2 below 2^-28)
// and, the planes of a truncated operand all having its sign, carry the sign of ab -- a bias towards zero of at most
// 2^-20 |ab| per product, about 2^-23 typically (the planes' leading bits are spread evenly), not zero-mean rounding noise.
// The weights that are split once per step (planes_pack_kernel, mlp_pack_kernel) are split round-to-nearest-even, which
// halves their side of it and makes it zero-mean; the activations are truncated in the kernels (2 VALU operations per
// plane instead of the 5 of an RNE split).  Measured against an fp64 matmul over the net's GEMM shapes the result is as
// close as the fp32 MFMA's (profiles/r03_x6_vs_fp32_mfma_gemm_probe.txt), at 6 x 32 instead of 8 x 64 matrix-pipe cycles
// per 32 x 32 x 16 block (v_mfma_f32_32x32x16_bf16 vs eight v_mfma_f32_32x32x2_f32).
typedef __bf16 vr_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned vr_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void vr_split3(const f32x4 lo4, const f32x4 hi4, vr_bf16x8 (&out)[3]) {
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
  vr_u32x4 q0, q1, q2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {      // v_perm_b32: {hi16(x[2e+1]), hi16(x[2e])}
    q0[e] = __builtin_amdgcn_perm(p0[2 * e + 1], p0[2 * e], 0x07060302u);
    q1[e] = __builtin_amdgcn_perm(p1[2 * e + 1], p1[2 * e], 0x07060302u);
    q2[e] = __builtin_amdgcn_perm(p2[2 * e + 1], p2[2 * e], 0x07060302u);
  }
  out[0] = __builtin_bit_cast(vr_bf16x8, q0);
  out[1] = __builtin_bit_cast(vr_bf16x8, q1);
  out[2] = __builtin_bit_cast(vr_bf16x8, q2);
}

// one bf16 value per operand, round-to-nearest-even (v_cvt_pk_bf16_f32): the "bf16 with MFMA conv path" of BASELINE configs[2..4]
__device__ __forceinline__ vr_bf16x8 vr_round8(const f32x4 lo4, const f32x4 hi4) {
  const vr_bf16x8 r = {(__bf16)lo4[0], (__bf16)lo4[1], (__bf16)lo4[2], (__bf16)lo4[3],
                       (__bf16)hi4[0], (__bf16)hi4[1], (__bf16)hi4[2], (__bf16)hi4[3]};
  return r;
}

__device__ __forceinline__ f32x16 vr_mfma_x6(const vr_bf16x8 (&a)[3], const vr_bf16x8 (&b)[3], f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);      // small terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

}  // namespace
