// Fused Mlp of a ClusterBlock (backbone/fusion/vr_coc.py:195-223, used at :264-271): fc1 -> GELU -> fc2 without the
// hidden activation making a round trip through HBM, forward and data gradient, gfx950.
//
//   forward   y[m, :]  = res[m, :] + ls * (W2 gelu(W1 x[m, :] + b1) + b2)          (optionally u = W1 x + b1 is stored)
//   backward  dx[m, :] = W1^T (gelu'(u[m, :]) * (W2^T (ls * dy[m, :])))             (stores du and h = gelu(u) for the
//                                                                                    two weight gradients)
// Both are the same program:  T = Wa X^T  (hidden x pixels, contraction over the C block channels),  an elementwise
// step on T,  Y^T += Wb T'  (channels x pixels, contraction over the hidden units), chunk of 32 hidden units by chunk.
// A workgroup owns 128 pixels (4 waves x 32); a wave keeps its 32 pixels' X operand in registers for the whole kernel
// (split once into three bf16 planes) and the C x 32 accumulators of Y^T.  The FIRST GEMM is computed transposed
// (hidden units on the accumulator rows, pixels on the lanes): its 32 x 32 accumulator then already IS the B operand
// of the second GEMM -- lane = pixel, 8 hidden units per lane and k16 step in the fixed order {0-3, 8-11} + 4 (lane / 32)
// -- so the hidden tile goes from accumulator to operand through registers only (GELU, split), never through LDS.
// The weights arrive pre-split ("x6", x6.h): vrnet_mlp_pack_f32 writes, per chunk, the three bf16 planes of the Wa
// rows and of the Wb columns in exactly the lane order of the MFMA A fragments (Wb with the hidden-unit permutation
// above), so a chunk is ONE contiguous block that goes global -> LDS by DMA and every fragment is one conflict-free
// ds_read_b128 -- no weight is split in the kernel (the six-product scheme spends most of its VALU time there).
// Two LDS stages: the DMA of chunk i + 1 is in flight while chunk i is computed.
#include "x6.h"

namespace {

struct MlpArgs {
  const float* x; long ldx;            // [M][C] rows: forward = normalised block input, backward = dy
  const float* xscale;                 // [C] or NULL: x[m, c] *= xscale[c] while loading (backward: the layer scale)
  const unsigned short* wpack;         // vrnet_mlp_pack_f32 planes for this direction
  const float* bias_a;                 // [HID] or NULL: added to T (forward: fc1.bias)
  float* upre; long ldu;               // forward: pre-activation output (NULL = not stored); backward: pre-activation input
  float* hout; long ldh;               // backward: h = gelu(u) output (operand of fc2's weight gradient)
  float* du; long lddu;                // backward: d pre-activation output (operand of fc1's weight gradient)
  float* y; long ldy;                  // [M][C] output
  const float* bias_b;                 // forward: fc2.bias
  const float* res; long ldres;        // forward: residual input
  const float* res_scale;              // forward: layer scale [C] (NULL = 1)
  double* stats; int stats_nb;         // forward: (sum, sumsq) of the stored outputs per 32 x 32 tile (igemm_common.h)
  const float* x2; long ldx2;          // backward with recomputed pre-activation (RC): the forward's input rows (normalised block input)
  int M, HID;
  int dbg;                             // diagnostic build only (VRNET_MLP_DBG): bit 0 = skip the hidden-sized stores, bit 1 =
                                       // counted wait that leaves the stores in flight (unsafe: timing experiments only)
};

constexpr int MLP_HID_MAX = 2560;

// NPL: 3 = fp32 products as six bf16 products (x6), 1 = operands rounded to bf16 (compute_dtype "bf16")
// HB (with NPL = 1): the hidden-sized tensors -- the pre-activation u, and in the backward pass h and du, the operands of the two
// weight gradients -- are bf16 in HBM (precision 4: compute_dtype "bf16" with bf16 tensors; the addresses in MlpArgs are then of
// bf16 elements, row strides in elements): half the bytes of the kernel's dominant traffic.
__device__ __forceinline__ f32x4 mlp_ld4(const float* base, long off, bool hb) {
  if (!hb) return *reinterpret_cast<const f32x4*>(base + off);
  const vr_bf16x4 v = *reinterpret_cast<const vr_bf16x4*>(reinterpret_cast<const unsigned short*>(base) + off);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void mlp_st4(float* base, long off, const f32x4 v, bool hb) {
  if (!hb) {
    *reinterpret_cast<f32x4*>(base + off) = v;
  } else {
    const vr_bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<vr_bf16x4*>(reinterpret_cast<unsigned short*>(base) + off) = b;
  }
}
// RC (backward only, round 5): the pre-activation u = W1 x + b1 is RECOMPUTED per chunk (one more GEMM against the forward's
// fc1 fragments, which ride in the same stage: [fc2^T | fc1^T | fc1]) instead of read -- the forward then stores no hidden-sized
// tensor at all and the backward reads one less (stage 0 at bs 8: 268 of 870 MB).  The recomputed accumulator IS the forward's
// (same fragments, same MFMA order), so with fp32 hidden tensors the results are the unfused kernel's bits.
template <int C, int MODE, int NPL, bool HB = false, bool RC = false>
__global__ __launch_bounds__(256, C <= 64 ? 2 : 1) void mlp_fused_kernel(const MlpArgs p) {
  static_assert(!RC || MODE == 1, "recompute is a backward variant");
  constexpr int KS = C / 16;                          // k16 steps of the first GEMM
  constexpr int CB = C / 32;                          // 32-channel row blocks of the second GEMM
  constexpr int A_BYTES = KS * NPL * 1024, B_BYTES = 2 * CB * NPL * 1024, ST_BYTES = A_BYTES + B_BYTES + (RC ? A_BYTES : 0);
  constexpr int NPIECE = ST_BYTES / 1024, PPW = NPIECE / 4;       // 1 KB DMA pieces per stage / per wave
  static_assert(NPIECE % 4 == 0, "whole pieces per wave");
  constexpr int BIAS_MAX = RC ? (C <= 64 ? 1024 : 1536) : MLP_HID_MAX;      // (RC: two workgroups of C = 64 must fit 160 KB of LDS)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ST_BYTES + BIAS_MAX * 4];
  float* bias_s = reinterpret_cast<float*>(smem + 2 * ST_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hf = lane >> 5;
  const int pix = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool live = pix < p.M;                        // M % 32 == 0: a wave is live or dead as a whole
  const long row = live ? pix : 0;
  const int nchunks = p.HID >> 5;

  if ((MODE == 0 || RC) && p.bias_a) {
    for (int i = tid; i < p.HID; i += 256) bias_s[i] = p.bias_a[i];
  }
  // ---- the wave's X operand: lane (pixel, hf) holds channels 16 ks + 8 hf .. + 7 of every k16 step
  vr_bf16x8 xs[KS][NPL];
  {
    const float* xr = p.x + row * p.ldx + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 lo = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
      f32x4 hi = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
      if (p.xscale) {
        lo *= *reinterpret_cast<const f32x4*>(p.xscale + 16 * ks + 8 * hf);
        hi *= *reinterpret_cast<const f32x4*>(p.xscale + 16 * ks + 8 * hf + 4);
      }
      if constexpr (NPL == 3) vr_split3(lo, hi, xs[ks]);
      else xs[ks][0] = vr_round8(lo, hi);
    }
  }
  vr_bf16x8 x2s[RC ? KS : 1][NPL];      // RC: the forward's X operand, for the recomputed first GEMM
  if constexpr (RC) {
    const float* xr = p.x2 + row * p.ldx2 + 8 * hf;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(xr + 16 * ks);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(xr + 16 * ks + 4);
      if constexpr (NPL == 3) vr_split3(lo, hi, x2s[ks]);
      else x2s[ks][0] = vr_round8(lo, hi);
    }
  }
  __syncthreads();        // bias copy visible; every ordinary load above has retired before the first DMA is counted

  const unsigned char* wp = reinterpret_cast<const unsigned char*>(p.wpack) + (long)lane * 16;
  auto issue = [&](int chunk) {
    const unsigned char* src = wp + (long)chunk * ST_BYTES + wave * (PPW * 1024);
    unsigned char* dst = smem + (chunk & 1) * ST_BYTES + wave * (PPW * 1024);
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
  };

  f32x16 Y[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[cb][r] = 0.f;

  // backward: the pre-activations of a chunk are requested one chunk ahead, together with that chunk's DMA
  f32x4 uin[4], unext[4];
  if (MODE == 1 && !RC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) uin[j] = mlp_ld4(p.upre, row * p.ldu + 8 * j + 4 * hf, HB);
  }
  issue(0);
  for (int hc = 0; hc < nchunks; ++hc) {
#ifdef VR_TUNING
    if (p.dbg & 2) {
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (hc + 1 < nchunks) {
      issue(hc + 1);
      if (MODE == 1 && !RC) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          unext[j] = mlp_ld4(p.upre, row * p.ldu + 32 * (hc + 1) + 8 * j + 4 * hf, HB);
      }
    }
    const unsigned char* st = smem + (hc & 1) * ST_BYTES + lane * 16;

    // ---- T = Wa X^T: 32 hidden units x 32 pixels
    f32x16 T;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      vr_bf16x8 a[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) a[pl] = *reinterpret_cast<const vr_bf16x8*>(st + (ks * NPL + pl) * 1024);
      if constexpr (NPL == 3) T = vr_mfma_x6(a, xs[ks], T);
      else T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], xs[ks][0], T, 0, 0, 0);
    }

    if constexpr (RC) {      // u = W1 x + b1 of this chunk, recomputed: the forward's first GEMM on the third region of the stage
      f32x16 U;
#pragma unroll
      for (int r = 0; r < 16; ++r) U[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        vr_bf16x8 a[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          a[pl] = *reinterpret_cast<const vr_bf16x8*>(st + A_BYTES + B_BYTES + (ks * NPL + pl) * 1024);
        if constexpr (NPL == 3) U = vr_mfma_x6(a, x2s[ks], U);
        else U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], x2s[ks][0], U, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uin[j] = f32x4{U[4 * j], U[4 * j + 1], U[4 * j + 2], U[4 * j + 3]};
        if (p.bias_a) uin[j] += *reinterpret_cast<const f32x4*>(bias_s + 32 * hc + 8 * j + 4 * hf);
      }
    }
    // ---- elementwise step.  Accumulator register 4 j + e of lane (pixel, hf) = hidden unit 32 hc + 8 j + 4 hf + e.
    f32x4 t4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t4[j] = f32x4{T[4 * j], T[4 * j + 1], T[4 * j + 2], T[4 * j + 3]};
    if (MODE == 0) {
      if (p.bias_a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] += *reinterpret_cast<const f32x4*>(bias_s + 32 * hc + 8 * j + 4 * hf);
      }
      if (p.upre && live && !(p.dbg & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mlp_st4(p.upre, row * p.ldu + 32 * hc + 8 * j + 4 * hf, t4[j], HB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[j][e] = vr_gelu(t4[j][e]);
    } else {
      f32x4 h4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float grad;
          h4[j][e] = vr_gelu_both(uin[j][e], grad);
          t4[j][e] *= grad;
        }
      if (live && !(p.dbg & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mlp_st4(p.hout, row * p.ldh + 32 * hc + 8 * j + 4 * hf, h4[j], HB);
          mlp_st4(p.du, row * p.lddu + 32 * hc + 8 * j + 4 * hf, t4[j], HB);
        }
      }
    }

    // ---- Y^T += Wb T': the accumulator registers 8 k2 .. 8 k2 + 7 are the B fragment of k16 step k2
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      vr_bf16x8 hs[NPL];
      if constexpr (NPL == 3) vr_split3(t4[2 * k2], t4[2 * k2 + 1], hs);
      else hs[0] = vr_round8(t4[2 * k2], t4[2 * k2 + 1]);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        vr_bf16x8 a[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          a[pl] = *reinterpret_cast<const vr_bf16x8*>(st + A_BYTES + ((k2 * CB + cb) * NPL + pl) * 1024);
        if constexpr (NPL == 3) Y[cb] = vr_mfma_x6(a, hs, Y[cb]);
        else Y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], hs[0], Y[cb], 0, 0, 0);
      }
    }
    if (MODE == 1 && !RC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) uin[j] = unext[j];
    }
  }

  // ---- epilogue: lane (pixel, hf), register 4 j + e of block cb = channel 32 cb + 8 j + 4 hf + e
  if (MODE == 0) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      double q1 = 0.0, q2 = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = 32 * cb + 8 * j + 4 * hf;
        f32x4 v = f32x4{Y[cb][4 * j], Y[cb][4 * j + 1], Y[cb][4 * j + 2], Y[cb][4 * j + 3]};
        if (p.bias_b) v += *reinterpret_cast<const f32x4*>(p.bias_b + c0);
        if (p.res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + row * p.ldres + c0);
          if (p.res_scale) v = rv + *reinterpret_cast<const f32x4*>(p.res_scale + c0) * v;
          else v = rv + v;
        }
        if (live) *reinterpret_cast<f32x4*>(p.y + row * p.ldy + c0) = v;
        if (p.stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double dv = (double)v[e];
            q1 += dv;
            q2 = __builtin_fma(dv, dv, q2);
          }
        }
      }
      if (p.stats) {        // statistics of exactly what was stored (fp64): GroupNorm of the consumer
        const double s1 = wave_sum(q1), s2 = wave_sum(q2);
        if (lane == 0 && live) {
          double* d = p.stats + ((long)(pix >> 5) * p.stats_nb + cb) * 2;
          d[0] = s1;
          d[1] = s2;
        }
      }
    }
  } else if (live) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(p.y + row * p.ldy + 32 * cb + 8 * j + 4 * hf) =
            f32x4{Y[cb][4 * j], Y[cb][4 * j + 1], Y[cb][4 * j + 2], Y[cb][4 * j + 3]};
  }
}

// ---- weight planes.  One thread per (chunk, fragment slot, lane): 8 source values -> 3 x 8 bf16.
// Slot s < KS: A fragment of k16 step s: Wa[32 hc + (lane & 31)][16 s + 8 (lane >> 5) + e].
// Slot KS + k2 * CB + cb: B-side fragment: Wb[32 cb + (lane & 31)][32 hc + 16 k2 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)].
// Round-to-nearest-even split: w = p0 + p1 + p2 exactly (each remainder has <= 16 resp. 8 significant bits), dropped
// cross terms of the six-product scheme <= 2^-24 |ab| each and zero-mean.
__device__ __forceinline__ unsigned short mlp_rne_bf16(float v) {
  const __bf16 b = (__bf16)v;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float mlp_bf16_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// wc != nullptr: a third region per chunk, the A fragments of wc (the recompute variant's fc1 rows): slots KS + 2 CB .. + KS - 1
__global__ void mlp_pack_kernel(const float* wa, long sa_r, long sa_c, const float* wb, long sb_r, long sb_c, int HID, int C,
                                int npl, unsigned short* out, const float* wc = nullptr, long sc_r = 0, long sc_c = 0) {
  const int KS = C / 16, CB = C / 32, SLOTS = KS + 2 * CB + (wc ? KS : 0);
  const long total = (long)(HID / 32) * SLOTS * 64;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int lane = t & 63;
  const long q = t >> 6;
  const int s = q % SLOTS, hc = q / SLOTS;
  const int r = lane & 31, hf = lane >> 5;
  float v[8];
  if (s < KS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wa[(long)(32 * hc + r) * sa_r + (long)(16 * s + 8 * hf + e) * sa_c];
  } else if (s >= KS + 2 * CB) {
    const int s3 = s - KS - 2 * CB;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wc[(long)(32 * hc + r) * sc_r + (long)(16 * s3 + 8 * hf + e) * sc_c];
  } else {
    const int k2 = (s - KS) / CB, cb = (s - KS) % CB;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = wb[(long)(32 * cb + r) * sb_r + (long)(32 * hc + 16 * k2 + (e & 3) + 8 * (e >> 2) + 4 * hf) * sb_c];
  }
  unsigned short* dst = out + ((long)hc * SLOTS + s) * npl * 512 + lane * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned short p0 = mlp_rne_bf16(v[e]);
    dst[e] = p0;
    if (npl == 3) {
      const float r1 = v[e] - mlp_bf16_f32(p0);
      const unsigned short p1 = mlp_rne_bf16(r1);
      const float r2 = r1 - mlp_bf16_f32(p1);
      dst[512 + e] = p1;
      dst[1024 + e] = mlp_rne_bf16(r2);
    }
  }
}

}  // namespace

/* Channel widths the fused Mlp kernels are built for (the stage-0 / stage-1 blocks of phi = l; 0 = none). */
extern "C" int vrnet_mlp_fused_ok(int C, int HID, long M) {
  return (C == 64 || C == 128) && HID % 32 == 0 && HID >= 32 && HID <= MLP_HID_MAX && M % 32 == 0 && M > 0;
}

extern "C" long vrnet_mlp_pack_bytes(int C, int HID, int precision) {
  const int npl = precision == 1 ? 1 : 3;
  return (long)(HID / 32) * (C / 16 + 2 * (C / 32)) * npl * 1024;
}

/* Weight planes of both directions in one call: fwd (Wa = fc1 [HID][C], Wb = fc2 [C][HID]) and bwd (Wa = fc2^T,
 * Wb = fc1^T); either output may be NULL. */
extern "C" int vrnet_mlp_pack_f32(const float* w1, const float* w2, int C, int HID, int precision, void* pack_fwd,
                                  void* pack_bwd, void* stream) {
  VR_CHECK_ARG(w1 && w2 && (pack_fwd || pack_bwd), "mlp_pack: null tensor");
  VR_CHECK_ARG(vrnet_mlp_fused_ok(C, HID, 32), "mlp_pack: no fused Mlp kernel for C = %d, hidden = %d", C, HID);
  VR_CHECK_ARG(precision == 1 || precision == 2 || precision == 4, "mlp_pack: precision 2 (x6) or 1 / 4 (bf16-rounded operands)");
  const int npl = precision == 2 ? 3 : 1;
  const long total = (long)(HID / 32) * (C / 16 + 2 * (C / 32)) * 64;
  hipStream_t st = vr_stream(stream);
  if (pack_fwd)
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, st, w1, (long)C, 1L, w2, (long)HID, 1L, HID, C,
                       npl, reinterpret_cast<unsigned short*>(pack_fwd));
  if (pack_bwd)
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, st, w2, 1L, (long)HID, w1, 1L, (long)C, HID, C,
                       npl, reinterpret_cast<unsigned short*>(pack_bwd));
  VR_LAUNCH_CHECK("mlp_pack");
  return VR_OK;
}

/* Whether the recompute form of the fused backward (vrnet_mlp_pack_rc_f32 / vrnet_mlp_bwd_rc_f32) exists: the fused kernels' shapes
 * with the hidden width its LDS budget allows (two workgroups of C = 64 must fit 160 KB). */
extern "C" int vrnet_mlp_rc_ok(int C, int HID, long M) { return vrnet_mlp_fused_ok(C, HID, M) && HID <= (C <= 64 ? 1024 : 1536); }

/* Pack of the backward kernel that recomputes the pre-activation (vrnet_mlp_bwd_rc_f32): per chunk [fc2^T | fc1^T | fc1]. */
extern "C" long vrnet_mlp_pack_rc_bytes(int C, int HID, int precision) {
  const int npl = precision == 2 ? 3 : 1;
  return (long)(HID / 32) * (2 * (C / 16) + 2 * (C / 32)) * npl * 1024;
}
extern "C" int vrnet_mlp_pack_rc_f32(const float* w1, const float* w2, int C, int HID, int precision, void* pack, void* stream) {
  VR_CHECK_ARG(w1 && w2 && pack, "mlp_pack_rc: null tensor");
  VR_CHECK_ARG(vrnet_mlp_rc_ok(C, HID, 32), "mlp_pack_rc: no recompute kernel for C = %d, hidden = %d", C, HID);
  VR_CHECK_ARG(precision == 1 || precision == 2 || precision == 4, "mlp_pack_rc: precision 2 (x6) or 1 / 4 (bf16-rounded operands)");
  const int npl = precision == 2 ? 3 : 1;
  const long total = (long)(HID / 32) * (2 * (C / 16) + 2 * (C / 32)) * 64;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(vr_cdiv(total, 256)), dim3(256), 0, vr_stream(stream), w2, 1L, (long)HID, w1, 1L, (long)C, HID, C,
                     npl, reinterpret_cast<unsigned short*>(pack), w1, (long)C, 1L);
  VR_LAUNCH_CHECK("mlp_pack_rc");
  return VR_OK;
}

template <int C>
static int mlp_launch(const MlpArgs& p, int mode, int precision, hipStream_t st) {
  const dim3 grid((unsigned)vr_cdiv(p.M, 128)), block(256);
  if (precision == 2) {
    if (mode == 0) hipLaunchKernelGGL((mlp_fused_kernel<C, 0, 3>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<C, 1, 3>), grid, block, 0, st, p);
  } else if (precision == 4) {
    if (mode == 0) hipLaunchKernelGGL((mlp_fused_kernel<C, 0, 1, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<C, 1, 1, true>), grid, block, 0, st, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((mlp_fused_kernel<C, 0, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<C, 1, 1>), grid, block, 0, st, p);
  }
  return VR_OK;
}

static bool mlp_vec_ok(const void* ptr, long ld) { return ptr == nullptr || (vr_aligned16(ptr) && ld % 4 == 0); }

extern "C" int vrnet_mlp_fwd_f32(const float* x, long ldx, const void* pack_fwd, const float* b1, const float* b2,
                                 const float* res, long ldres, const float* res_scale, float* y, long ldy, float* upre,
                                 long ldu, double* stats, long M, int C, int HID, int precision, void* stream) {
  VR_CHECK_ARG(x && pack_fwd && y, "mlp_fwd: null tensor");
  VR_CHECK_ARG(vrnet_mlp_fused_ok(C, HID, M), "mlp_fwd: no fused Mlp kernel for C = %d, hidden = %d, %ld rows", C, HID, M);
  VR_CHECK_ARG(precision == 1 || precision == 2 || precision == 4, "mlp_fwd: precision 2 (x6), 1 (bf16-rounded operands) or 4 (1 with "
                                                                      "the pre-activation stored as bf16)");
  VR_CHECK_ARG(M < (1L << 31), "mlp_fwd: too many rows");
  VR_CHECK_ARG(mlp_vec_ok(x, ldx) && mlp_vec_ok(y, ldy) && mlp_vec_ok(res, ldres) && mlp_vec_ok(upre, ldu) &&
                   mlp_vec_ok(b1, 0) && mlp_vec_ok(b2, 0) && mlp_vec_ok(res_scale, 0) && vr_aligned16(pack_fwd),
               "mlp_fwd: tensors must be 16-byte aligned with row strides that are multiples of 4");
  VR_CHECK_ARG(ldx >= C && ldy >= C && (!res || ldres >= C) && (!upre || ldu >= HID), "mlp_fwd: row stride smaller than width");
  if (vr_ablated("igemm") || vr_ablated("igemm_big")) return VR_OK;
  MlpArgs p{};
  p.x = x; p.ldx = ldx; p.wpack = reinterpret_cast<const unsigned short*>(pack_fwd); p.bias_a = b1;
  p.upre = upre; p.ldu = ldu; p.y = y; p.ldy = ldy; p.bias_b = b2; p.res = res; p.ldres = ldres; p.res_scale = res_scale;
  p.stats = stats; p.stats_nb = C / 32; p.M = (int)M; p.HID = HID;
  p.dbg = vr_tune("VRNET_MLP_DBG", 0);
  if (C == 64) mlp_launch<64>(p, 0, precision, vr_stream(stream));
  else mlp_launch<128>(p, 0, precision, vr_stream(stream));
  vr_note_kernel(precision == 2 ? 7 : 8);
  VR_LAUNCH_CHECK("mlp_fwd");
  return VR_OK;
}

extern "C" int vrnet_mlp_bwd_f32(const float* dy, long lddy, const float* dy_scale, const void* pack_bwd, const float* upre,
                                 long ldu, float* h, long ldh, float* du, long lddu, float* dx, long lddx, long M, int C,
                                 int HID, int precision, void* stream) {
  VR_CHECK_ARG(dy && pack_bwd && upre && h && du && dx, "mlp_bwd: null tensor");
  VR_CHECK_ARG(vrnet_mlp_fused_ok(C, HID, M), "mlp_bwd: no fused Mlp kernel for C = %d, hidden = %d, %ld rows", C, HID, M);
  VR_CHECK_ARG(precision == 1 || precision == 2 || precision == 4, "mlp_bwd: precision 2 (x6), 1 (bf16-rounded operands) or 4 (1 with "
                                                                      "upre / h / du as bf16 tensors)");
  VR_CHECK_ARG(M < (1L << 31), "mlp_bwd: too many rows");
  VR_CHECK_ARG(mlp_vec_ok(dy, lddy) && mlp_vec_ok(upre, ldu) && mlp_vec_ok(h, ldh) && mlp_vec_ok(du, lddu) &&
                   mlp_vec_ok(dx, lddx) && mlp_vec_ok(dy_scale, 0) && vr_aligned16(pack_bwd),
               "mlp_bwd: tensors must be 16-byte aligned with row strides that are multiples of 4");
  VR_CHECK_ARG(lddy >= C && lddx >= C && ldu >= HID && ldh >= HID && lddu >= HID, "mlp_bwd: row stride smaller than width");
  if (vr_ablated("igemm") || vr_ablated("igemm_big")) return VR_OK;
  MlpArgs p{};
  p.x = dy; p.ldx = lddy; p.xscale = dy_scale; p.wpack = reinterpret_cast<const unsigned short*>(pack_bwd);
  p.upre = const_cast<float*>(upre); p.ldu = ldu; p.hout = h; p.ldh = ldh; p.du = du; p.lddu = lddu; p.y = dx; p.ldy = lddx;
  p.M = (int)M; p.HID = HID;
  p.dbg = vr_tune("VRNET_MLP_DBG", 0);
  if (C == 64) mlp_launch<64>(p, 1, precision, vr_stream(stream));
  else mlp_launch<128>(p, 1, precision, vr_stream(stream));
  vr_note_kernel(precision == 2 ? 7 : 8);
  VR_LAUNCH_CHECK("mlp_bwd");
  return VR_OK;
}

/* vrnet_mlp_bwd_f32 WITHOUT the stored pre-activation (round 5): u = W1 x + b1 is recomputed chunk by chunk from the forward's
 * input rows x (the normalised block input; fp32, row stride ldx) against pack = vrnet_mlp_pack_rc_f32's [fc2^T | fc1^T | fc1]
 * -- the forward pass (vrnet_mlp_fwd_f32 with upre = NULL) then writes no hidden-sized tensor and this kernel reads one less.
 * precision 2: h, du fp32 -- bit-identical to vrnet_mlp_bwd_f32 on the stored u; 4: h, du bf16 tensors (u itself stays fp32-
 * accurate here, where precision 4 of the stored form rounded it to bf16).  HID <= 1024 (C = 64) / 1536 (C = 128). */
extern "C" int vrnet_mlp_bwd_rc_f32(const float* dy, long lddy, const float* dy_scale, const void* pack, const float* x, long ldx,
                                    const float* b1, float* h, long ldh, float* du, long lddu, float* dx, long lddx, long M, int C,
                                    int HID, int precision, void* stream) {
  VR_CHECK_ARG(dy && pack && x && h && du && dx, "mlp_bwd_rc: null tensor");
  VR_CHECK_ARG(vrnet_mlp_rc_ok(C, HID, M),
               "mlp_bwd_rc: no recompute kernel for C = %d, hidden = %d, %ld rows", C, HID, M);
  VR_CHECK_ARG(precision == 2 || precision == 4, "mlp_bwd_rc: precision 2 (x6, fp32 h / du) or 4 (bf16-rounded operands, bf16 h / du)");
  VR_CHECK_ARG(M < (1L << 31), "mlp_bwd_rc: too many rows");
  VR_CHECK_ARG(mlp_vec_ok(dy, lddy) && mlp_vec_ok(x, ldx) && mlp_vec_ok(h, ldh) && mlp_vec_ok(du, lddu) && mlp_vec_ok(dx, lddx) &&
                   mlp_vec_ok(dy_scale, 0) && mlp_vec_ok(b1, 0) && vr_aligned16(pack),
               "mlp_bwd_rc: tensors must be 16-byte aligned with row strides that are multiples of 4");
  VR_CHECK_ARG(lddy >= C && lddx >= C && ldx >= C && ldh >= HID && lddu >= HID, "mlp_bwd_rc: row stride smaller than width");
  if (vr_ablated("igemm") || vr_ablated("igemm_big")) return VR_OK;
  MlpArgs p{};
  p.x = dy; p.ldx = lddy; p.xscale = dy_scale; p.wpack = reinterpret_cast<const unsigned short*>(pack);
  p.x2 = x; p.ldx2 = ldx; p.bias_a = b1;
  p.hout = h; p.ldh = ldh; p.du = du; p.lddu = lddu; p.y = dx; p.ldy = lddx;
  p.M = (int)M; p.HID = HID;
  const dim3 grid((unsigned)vr_cdiv(M, 128)), block(256);
  hipStream_t st = vr_stream(stream);
  if (C == 64) {
    if (precision == 2) hipLaunchKernelGGL((mlp_fused_kernel<64, 1, 3, false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<64, 1, 1, true, true>), grid, block, 0, st, p);
  } else {
    if (precision == 2) hipLaunchKernelGGL((mlp_fused_kernel<128, 1, 3, false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<128, 1, 1, true, true>), grid, block, 0, st, p);
  }
  vr_note_kernel(precision == 2 ? 7 : 8);
  VR_LAUNCH_CHECK("mlp_bwd_rc");
  return VR_OK;
}
