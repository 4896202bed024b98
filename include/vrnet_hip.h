/* vrnet_hip.h -- C ABI of libvrnet_hip.so: MI355X (gfx950) kernels for the ASY-VRNet fusion hot path.
 *
 * The drop-in boundary of this repository (DESIGN.md "Boundary").  Plain pointers and sizes only: no
 * torch types.  Every pointer is a DEVICE pointer unless stated; every function enqueues work on the HIP
 * stream `stream` (a hipStream_t passed as void*) and returns without synchronising:
 *   0 = ok, 1 = bad argument, 2 = launch failure, 3 = workspace too small; the message of the last
 *   failure on the calling thread is returned by vrnet_last_error().
 * Tensors are fp32, NHWC, addressed as rows of pixels: element (pixel r, channel c) of tensor t lives at
 * t[r * ld + c]; `ld` (row stride in floats) >= channel count lets a tensor be a channel slice of a wider
 * buffer (torch.cat is never materialised by a copy of the producer's output).  `accumulate` != 0 makes an
 * output `+=` instead of `=` (gradient fan-in).
 *
 * Each entry point names the reference code (GuanRunwei/ASY-VRNet, file:line) it replaces; the reference
 * is pure PyTorch, so "replaces" means the ATen op sequence the reference module executes there.
 * The reference-side binding a maintainer would add is a ctypes stub: see INTEGRATION.md.
 */
#ifndef VRNET_HIP_H
#define VRNET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------------- */
int vrnet_abi_version(void);                 /* == 10 */
/* Kernel family the last vrnet_conv2d_f32 / vrnet_conv2d_wgrad_f32 call of this thread dispatched to: 1 fp32 MFMA
 * (register-staged), 2 fp32 MFMA (LDS-DMA ring), 3 bf16-rounded operands, 4 direct kernels for tiny channel counts,
 * 5 direct HBM-streaming kernels for 1x1 convs with <= 16 output channels over wide inputs (head predictions, seg logits),
 * 6 "x6": every fp32 product as six exact bf16 x bf16 products on the bf16 MFMA, fp32 accumulate;
 * 9: x6 with pre-split weights (vrnet_conv2d_f32 `w_planes`);
 * 7 / 8: the fused Mlp kernels (vrnet_mlp_fwd_f32 / vrnet_mlp_bwd_f32) at precision 2 (x6) / 1 (bf16-rounded operands);
 * 10 / 11: plane GEMM forward / data gradient (vrnet_gemm_planes_f32) with np = 3 / 1; 12 / 13: plane weight gradient. */
int vrnet_last_kernel(void);
/* Launches of kernel family `family` (same codes) issued by this process (all threads) since the library was loaded. */
long vrnet_kernel_launches(int family);
const char* vrnet_last_error(void);          /* host string, thread local */
/* Diagnostic: a one-thread kernel writes the device's constant-rate clock (wall_clock64, 100 MHz) to *dst in stream order:
 * section timelines of the captured step without a tracer attached (no reference counterpart). */
int vrnet_clock_stamp(long long* dst, void* stream);
/* 0 for the product library, which reads NO environment variable; 1 for the diagnostic build (make tuning:
 * libvrnet_hip_tuning.so, -DVR_TUNING) in which the VRNET_* dispatch knobs and the VRNET_ABLATE launch-skipping timing
 * ablation exist.  Benchmarks must refuse a library that answers 1. */
int vrnet_tuning_build(void);
int vrnet_device_arch(char* buf, int len);   /* host buffer <- e.g. "gfx950:sramecc+:xnack-" (synchronous) */

/* ---- dense convolution as implicit GEMM on the fp32 MFMA ------------------------------------------
 * Replaces nn.Conv2d forward / input-gradient for every dense conv of the path:
 *   Cluster fc1, fc_v, fc2 (backbone/fusion/vr_coc.py:145-147,156-157,191), Mlp fc1/fc2 (:205-207,217-223),
 *   PointRecuder.proj k1 / k4s4 / k3s2p1 (:83-102), BaseConv.conv (backbone/conv_utils/normal_conv.py:41,48),
 *   ASPP 1x1 + dilated 3x3 (neck/coc_fpn_dual.py:50-77), head stems / pconv / preds (head/decouplehead.py:21-40).
 * Geometry is always the FORWARD convolution's: input (B,H,W,Cin), output (B,OH,OW,Cout), kernel kh x kw.
 * w: weights packed [kh*kw][Cout][Cin] (vrnet_pack_weight_f32; for 1x1 this IS the OIHW tensor).
 * mode 0: y(B,OH,OW,Cout) = conv(a = x).   mode 1: y(B,H,W,Cin) = d/dx given a = dy(B,OH,OW,Cout).
 * Fused epilogue, in this order (each optional, NULL/0 = off):
 *   v = acc + bias[n];  v *= gelu'(aux[m,n]) (Mlp backward);  ypre[m,n] = v (pre-activation / branch output);
 *   act: 1 ReLU, 2 exact-erf GELU;  v = res[m,n] + res_scale[n] * v (layer-scale residual, vr_coc.py:266-271);
 *   store NHWC y[m*ldy+n] or NCHW y[b][out_coff+n][pix] of a (B,out_ctot,OH,OW) tensor (head cat, decouplehead.py:86).
 * kscale[k]: multiplies the contraction channels of `a` (layer scale folded into the data gradient).
 * precision: 2 = every fp32 product as six exact bf16 x bf16 products (operands split into three bf16 values each,
 *   a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1 on v_mfma_f32_32x32x16_bf16, fp32 accumulate: fp32 rounding-level error,
 *   2.7x the matrix rate of the fp32 MFMA) on the LDS-DMA tile kernels where vrnet_conv2d_dma_tile() != 0, the fp32 MFMA
 *   elsewhere -- the default of the fp32 path; 3 = bf16-rounded operands on those tile kernels, standard weight layout;
 *   0 = fp32 operands on v_mfma_f32_32x32x2_f32 only; 1 = operands rounded to bf16 while staged,
 *   v_mfma_f32_32x32x16_bf16, fp32 accumulate and epilogue (BASELINE configs "bf16 with MFMA conv path"): needs
 *   16-byte rows, Cin % 4 == 0 (mode 0) / Cout % 4 == 0 (mode 1), > 32 GEMM columns, and in mode 1 `w` =
 *   vrnet_pack_weight_t_f32's [kh*kw][Cin][Cout] pack with kscale folded in (kscale itself must then be NULL).
 * stats (NULL = none; forward, NHWC, Cout > 32, OH*OW % 32 == 0): [ceil(M/32)][ceil(Cout/32)][2] fp64 (sum, sum of
 *   squares) of the STORED outputs per 32-row x 32-channel tile -- the GroupNorm statistics of the consumer
 *   (vrnet_gn_coef_from_pairs: OH*OW/32 * ceil(Cout/32) consecutive pairs per sample) without another pass over y.
 * Two-stream launch (pair_rows > 0): the image chain and the radar chain of a backbone stage run the same layer
 *   shapes on their own parameters (vr_coc.py:589-600: network[idx](x), network_radar[idx](x_radar)); with both
 *   streams stacked along the batch, GEMM rows < pair_rows use (w, bias, res_scale, kscale) and rows >= pair_rows use
 *   (w2, bias2, res_scale2, kscale2) -- one launch with twice the tiles instead of two.  pair_rows % 128 == 0. */
/* Tile the LDS-DMA x6 / bf16 kernels would use for a GEMM of `rows` x `cols` (mode 0: output pixels x Cout; mode 1: input
 * pixels x Cin): 22 (128 x 128), 21 (128 x 64) or 0 = no such kernel for the shape.  precision 2 falls back to the fp32
 * MFMA by itself; precision 3 (bf16-rounded operands with the STANDARD weight layout in both modes, kscale allowed) is only
 * accepted where this returns non-zero. */
int vrnet_conv2d_dma_tile(long rows, int cols);
/* Optional column statistics of the STORED outputs (NULL = none; NHWC vector epilogue, > 32 output channels, not for
 * stride-2 data gradients): what a separate pass over the output tensor would otherwise compute.
 *   partial[mb][n] = (sum_m v[m,n], sum_m v[m,n] * f[m,n]) over the 32 rows of row tile mb, f = x2 (row stride ldx2) or,
 *     with x2 == NULL, v itself: the chunk partials of train-mode BatchNorm statistics (vrnet_bn_coef_fwd_partial, forward
 *     conv of a BaseConv, normal_conv.py:45-49) and of GroupNorm's backward moments (data gradient of the conv behind a
 *     GroupNorm, x2 = the GroupNorm input; vr_coc.py:264-271), in the layout [row tile][channel][2] of the moments kernel;
 *   tile_totals[mb][n / 32] = the same two sums weighted by gamma[n] and added over the tile's 32 columns (GroupNorm
 *     backward: per-sample coefficients, vrnet_gn_apply_bwd_from_partials). */
typedef struct vrnet_conv_colstats {
  double* partial;
  const float* x2; long ldx2;
  const float* gamma;
  double* tile_totals;
} vrnet_conv_colstats;
int vrnet_conv2d_f32(const float* a, long lda, const float* w, const float* bias, float* y, long ldy,
                     int B, int H, int W, int Cin, int OH, int OW, int Cout, int kh, int kw, int stride, int pad,
                     int dil, int mode, int act, float* ypre, long ldypre, const float* res, long ldres,
                     const float* res_scale, const float* kscale, const float* aux, long ldaux, int out_nchw,
                     int out_ctot, int out_coff, int accumulate, double* stats, int precision, int pair_rows,
                     const float* w2, const float* bias2, const float* res_scale2, const float* kscale2,
                     const void* w_planes, const vrnet_conv_colstats* colstats, void* workspace, long workspace_bytes,
                     void* stream);
/* Split contraction (round 4): a 2048-row map (16 x 16 at batch 8) has 64-128 tiles of 128 x 64 for 256 CUs, so at
 * precision 2 such layers ran on 64 x 64 fp32-MFMA tiles at 40-70 TFLOP/s.  With a `workspace` of
 * vrnet_conv2d_splitk_workspace(rows, cols, ktot) bytes (rows x cols = the GEMM: B*OH*OW x Cout forward, B*H*W x Cin data
 * gradient; ktot = contraction length incl. taps; 0 = the shape is not split) the x6 tile kernels take them: `splits`
 * workgroups per tile each run a share of the K loop and leave raw accumulators in the workspace, a finishing launch adds
 * the slabs in order (deterministic) and runs the usual fused epilogue.  workspace NULL: never split.
 * vrnet_conv2d_dma_plan: the tile (as vrnet_conv2d_dma_tile, or 21 for a split launch) and *splits. */
long vrnet_conv2d_splitk_workspace(long rows, int cols, long ktot);
int vrnet_conv2d_dma_plan(long rows, int cols, long ktot, int* splits);
/* Pre-split weights for the x6 kernels (precision 2, 1x1 convs, contraction % 16 == 0): the six-product scheme spends its
 * VALU time on splitting fragments into bf16 planes; weights are the same for every row tile of a step, so they can be
 * split ONCE per step.  vrnet_conv_planes_pack_f32 does that for a whole table of weights in one launch (round-to-nearest-
 * even: w = p0 + p1 + p2 exactly), writing each as the LDS stage image of the kernel ([k16 step][64-column block][plane]
 * [lane half][64 columns] x 8 bf16); `w_planes` of vrnet_conv2d_f32 passes one such pack (NULL = split in the kernel).
 * mode 0: columns J = Cout, contraction K = Cin (source strides Cin, 1 for an OIHW 1x1 weight); mode 1: J = Cin, K = Cout
 * (strides 1, Cin) with kscale folded in through the table's scale address (the kscale argument is then ignored).
 * `w` is still required: shapes without a tile kernel (vrnet_conv2d_dma_tile == 0) use it.  Kernel family 9. */
long vrnet_conv_planes_bytes(int J, int K);
int vrnet_conv_planes_pack_f32(const long* table, int nentries, long total_blocks, void* stream);

/* ---- Plane GEMMs (round 4, csrc/pgemm.hip): the 1x1 convolutions of the ClusterBlocks (fc1 | fc_v, the Cluster's proj, the
 * Mlp's fc1 / fc2: backbone/fusion/vr_coc.py:145-147, 187, 205-207, neck variant backbone/vision/context_cluster.py:211-216)
 * and their data gradients on operands that ALREADY ARE bf16 planes in HBM, so that the GEMM main loop is LDS-DMA +
 * ds_read_b128 + MFMA with no VALU work on operands.
 *   plane tensor: element (r, k) of plane q at base[q * plane + r * ld + k], bf16; ld and plane multiples of 8, base 16-byte
 *   aligned.  np = 3: an fp32 tensor t split EXACTLY, t = p0 + p1 + p2 (round-to-nearest-even at each step; written by the
 *   producing kernel's epilogue, by vrnet_planes_split_f32 for the weights of a step, or by vrnet_planes_from_f32); products
 *   a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1 accumulated in fp32 on v_mfma_f32_32x32x16_bf16 -- the x6 scheme of precision 2
 *   with the same error bound.  np = 1: bf16 tensors, one product (compute_dtype "bf16": bf16 activations in HBM).
 * vrnet_gemm_planes_f32: y[M][N] = epilogue(A[M][K] . B[N][K]^T); forward: A = activations, B = w[Cout][Cin]; data gradient:
 * A = dy, B = transposed weights with the layer scale folded in.  Epilogue as vrnet_conv2d_f32: bias, aux (x gelu'(aux)),
 * ypre, act (0 / 1 ReLU / 2 GELU), res (+ res_scale), accumulate, stats (fp64 (sum, sumsq) per 32 x 32 tile; stats_hw = rows
 * per sample), colstats.  The result goes to `y` (fp32; may be NULL) and / or to `yp` as planes (yp_np 3 or 1).
 * K % 32 == 0, N % 4 == 0; ask vrnet_gemm_planes_ok for shapes with too few 128 x 128 tiles.  half_side: bit 0 = `ypre` is a bf16
 * tensor (stored rounded), bit 1 = `aux` is a bf16 tensor (row strides in elements): the Mlp's pre-activation u in bf16 mode.
 * Kernel families 10 (np 3), 11 (np 1). */
int vrnet_gemm_planes_ok(long rows, int cols, int K);
int vrnet_gemm_planes_f32(const void* a, long lda, long a_plane, const void* b, long ldb, long b_plane, int np, long M, int N,
                          int K, const float* bias, float* y, long ldy, void* yp, long ldyp, long yp_plane, int yp_np, int act,
                          float* ypre, long ldypre, const float* res, long ldres, const float* res_scale, const float* aux,
                          long ldaux, int accumulate, double* stats, long stats_hw, const vrnet_conv_colstats* colstats,
                          int half_side, void* stream);
/* fp32 matrices -> planes in one launch for a table of matrices (the weights of a step).  Entry (10 longs): source address,
 * rows R, contraction K, source element strides (row, k), address of a scale per k or 0 (the layer scale of a data-gradient
 * pack), destination address, destination row stride, destination plane stride, first block (running sum of
 * vrnet_planes_split_blocks). */
long vrnet_planes_split_blocks(long R, long K);
int vrnet_planes_split_f32(const long* table, int nentries, long total_blocks, int np, void* stream);
/* One row-major fp32 matrix (R rows of K values, row stride lds) -> planes: conversion pass for an activation tensor whose
 * producer has no plane output. */
int vrnet_planes_from_f32(const float* src, long lds, long R, long K, void* dst, long ld, long plane, int np, void* stream);
/* Weight (+ bias, + layer-scale) gradient of a 1x1 conv on plane operands: dw[Cout][Cin] (+)= row_scale[n] * sum_m dy[m][n] *
 * x[m][c]; dbias, row_scale, accumulate, (w, bias, dls) as vrnet_conv2d_wgrad_f32 (autograd of vr_coc.py:145-147, 187, 205-207).
 * x: planes of the conv's input (M rows of Cin), dy: planes of the output gradient (M rows of Cout), same np.  The fragments
 * -- 8 consecutive rows of one column -- are read with ds_read_b64_tr_b16; no operand is split in the kernel (the in-kernel
 * split weight gradient splits four fragments per 24 MFMAs).  Channel counts multiples of 8.  Families 12 (np 3) / 13 (np 1). */
/* Optional bf16-plane output of a producing kernel: the tensor it hands to a plane GEMM.  plane q of element (row, c) at
 * p[q * plane + row * ld + c]; np = 3 exact split, 1 rounded to bf16. */
typedef struct vrnet_planes_out {
  void* p;
  long ld, plane;
  int np;
} vrnet_planes_out;
/* Producers with a plane output (same arguments as the functions they extend, plus the planes):
 *   vrnet_gn_apply_fwd_planes    GroupNorm(1, C) output (vr_coc.py:264, 268) -- y may be NULL (planes only);
 *   vrnet_gn_apply_bwd_planes    its input gradient: the block's outgoing gradient as the next GEMMs' dy operand;
 *   vrnet_cluster_fwd_planes_f32 the Cluster core's output (vr_coc.py:158-186), forced != 0 = the teacher-forced form; the fp32
 *     `out` may be NULL (planes only); in_bf16 != 0: f and v are bf16 tensors (ld in elements) -- what the reference's
 *     autocast hands its Cluster (train.py:345-350): the fc1 | fc_v GEMM then writes 2 bytes per element and nothing else;
 *   vrnet_cluster_bwd_planes_f32 [df | dv] as ONE plane tensor of 2 E D columns (df first); df / dv may both be NULL (planes
 *     only); in_bf16 != 0: f, v AND dout are bf16 tensors. */
int vrnet_gn_apply_fwd_planes(const float* x, long ldx, const double* pairs, long pairs_per_sample, const float* gamma,
                              const float* beta, float eps, int B, long HW, int C, float* y, long ldy, float* mean_rstd,
                              const vrnet_planes_out* yp, void* stream);
int vrnet_gn_apply_bwd_planes(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd, const float* gamma,
                              int B, long HW, int C, const float* add, long ldadd, float* out, long ldo, float* dgamma,
                              float* dbeta, int accumulate_params, const vrnet_planes_out* outp, void* workspace,
                              long workspace_bytes, void* stream);
/* Regions of more than 256 points (every backbone stage at 1 024 px, neck p3 at 512 px) run on a kernel that streams the
 * points in chunks; its backward re-read f four times and v three times.  `state` (vrnet_cluster_state_floats(...) floats, 0 for
 * the register-resident region sizes; NULL = not wanted): the forward leaves the region centres, aggregated values and
 * assignment counts there; the backward, given that state and the forward's similarity map `wgt_fwd` (both or neither), starts
 * at its third pass -- f twice, v once, the same bits (outp / dfvp may be NULL in both: no plane copies). */
long vrnet_cluster_state_floats(int B, int H, int W, int E, int fold);
int vrnet_cluster_fwd_planes_f32(const void* f, const void* v, long ld, int in_bf16, const float* alpha, const float* beta,
                                 float* out, long ldo, unsigned char* idx, float* wgt, int B, int H, int W, int E, int D,
                                 int fold, int forced, const vrnet_planes_out* outp, float* state, void* stream);
int vrnet_cluster_bwd_planes_f32(const void* f, const void* v, long ld, int in_bf16, const float* alpha, const float* beta,
                                 const unsigned char* idx, const void* dout, long lddo, float* df, float* dv, long lddf,
                                 float* dalpha, float* dbeta, int accumulate_ab, int B, int H, int W, int E, int D, int fold,
                                 const vrnet_planes_out* dfvp, const float* wgt_fwd, const float* state, void* workspace,
                                 long workspace_bytes, void* stream);
int vrnet_wgrad_planes_ok(long M, int Cin, int Cout);
long vrnet_wgrad_planes_workspace(long M, int Cin, int Cout);
int vrnet_wgrad_planes_f32(const void* x, long ldx, long x_plane, const void* dy, long lddy, long dy_plane, int np, long M, int Cin,
                           int Cout, float* dw, float* dbias, const float* row_scale, int accumulate, const float* w,
                           const float* bias, float* dls, void* workspace, long workspace_bytes, void* stream);

/* Weight (+ bias) gradient of the same convolutions (autograd of nn.Conv2d): dw in OIHW layout
 * [Cout][Cin][kh][kw], dbias[Cout] (NULL = none), both scaled by row_scale[Cout] when given (layer scale).
 * Deterministic split over output pixels into fp32 slabs in `workspace` (size from ..._workspace).
 * precision 1: dy and x rounded to bf16 while staged, v_mfma_f32_32x32x16_bf16 with transposing LDS reads, fp32
 * accumulate / slabs / bias sums (needs 16-byte rows, Cin, Cout multiples of 4 and > 32).
 * Two-stream launch (dw2 != NULL; workspace with pair = 1): samples [0, B/2) contribute to (dw, dbias, row_scale),
 * samples [B/2, B) to (dw2, dbias2, row_scale2).
 * dls (NULL = none; 1x1 convs): gradient of the layer scale behind this conv (vr_coc.py:266-271, x + ls * conv(h)):
 *   dls[n] (+)= sum_c w[n][c] * dw_raw[n][c] + bias[n] * db_raw[n]  (raw = before row_scale), which equals
 *   sum_m dy[m,n] * conv(h)[m,n] -- so the branch output is neither stored in the forward pass nor re-read. */
long vrnet_conv2d_wgrad_workspace(int B, int OH, int OW, int Cin, int Cout, int kh, int kw, int pair);
int vrnet_conv2d_wgrad_f32(const float* x, long ldx, const float* dy, long lddy, float* dw, float* dbias,
                           const float* row_scale, int B, int H, int W, int Cin, int OH, int OW, int Cout, int kh,
                           int kw, int stride, int pad, int dil, int accumulate, int precision, float* dw2,
                           float* dbias2, const float* row_scale2, const float* w, const float* bias, float* dls,
                           const float* w2, const float* bias2, float* dls2, void* workspace, long workspace_bytes,
                           void* stream);
int vrnet_pack_weight_f32(const float* w_oihw, float* w_tnc, int Cout, int Cin, int kh, int kw, void* stream);
/* [kh*kw][Cin][Cout] = w_oihw[n][c][t] * kscale[n] (kscale NULL = 1): the data-gradient operand of the bf16 path. */
int vrnet_pack_weight_t_f32(const float* w_oihw, const float* kscale, float* w_tcn, int Cout, int Cin, int kh, int kw,
                            void* stream);

/* ---- fused Mlp: fc1 -> GELU -> fc2 in ONE kernel per direction ---------------------------------------
 * Replaces Mlp.forward (backbone/fusion/vr_coc.py:217-223; neck variant backbone/vision/context_cluster.py) together with
 * the layer-scale residual around it (vr_coc.py:270-271: x + layer_scale_2 * mlp(norm2(x))) and their autograd: the
 * hidden activation (8x / 4x the block width) never makes a round trip through HBM.  Rows are pixels (M = B*H*W),
 * C = block width, HID = hidden width; kernels exist for C in {64, 128}, HID % 32 == 0, HID <= 2560, M % 32 == 0
 * (vrnet_mlp_fused_ok; callers keep the two vrnet_conv2d_f32 launches elsewhere).
 * Weights are consumed as bf16 PLANES in MFMA fragment order, written once per step by vrnet_mlp_pack_f32 from the
 * state_dict tensors w1 = fc1.weight [HID][C], w2 = fc2.weight [C][HID] (1x1 OIHW): precision 2 = three planes per weight,
 * w = p0 + p1 + p2 exactly (round-to-nearest-even splits), and every fp32 product is evaluated as the six bf16 x bf16
 * products p_i q_j with i + j <= 2 (exact in fp32), fp32 accumulate -- the "x6" arithmetic of vrnet_conv2d_f32 precision
 * 2; precision 1 = one plane, operands rounded to bf16.  Activations are split / rounded in registers.  vrnet_mlp_pack_bytes
 * = size of ONE direction's planes.
 * forward:  u = x w1^T + b1 (stored to upre when non-NULL: the backward pass needs it);  y = res + res_scale * (gelu(u) w2^T
 *   + b2)  (exact-erf GELU; res / res_scale / b1 / b2 optional);  stats as in vrnet_conv2d_f32 ([M/32][C/32][2] fp64).
 * backward: given dy and the stored u:  du = gelu'(u) * ((dy * dy_scale) w2);  dx = du w1;  h = gelu(u) is recomputed.
 *   du and h are written once (operands of the two weight gradients, vrnet_conv2d_wgrad_f32), dx is the data gradient.
 * precision 4 (round 4): precision 1 with the hidden-sized tensors in bf16 -- `upre` (forward: written, backward: read), `h` and
 *   `du` are then addresses of bf16 elements (row strides in elements): half the bytes of the kernels' dominant traffic; h and
 *   du are what vrnet_wgrad_planes_f32 (np = 1) takes as they are.  The forward output does not change (u is rounded only on
 *   its way to memory); the backward pass evaluates gelu / gelu' at the rounded u.  Packs: those of precision 1.
 * x6 on non-finite / tiny operands (both here and in vrnet_conv2d_f32 precision 2): an operand of +-Inf splits into
 *   (Inf, NaN, NaN), so an Inf in the data yields NaN where fp32 arithmetic yields Inf; NaN stays NaN.  The low planes of an
 *   operand below 2^-110 in magnitude fall into the bf16 denormal range, which the matrix pipe flushes: such operands
 *   carry 8-16 instead of 24 significant bits (the product is below 2^-110 |other operand|).
 * x6 error: the three dropped products are each below 2^-21 |ab| (typically 2^-23); the activation operands are split by
 *   truncation in the kernels, so their share is a bias towards zero rather than zero-mean rounding; the pre-split
 *   weights round to nearest even.  Against an fp64 matmul the results are as close as the fp32 MFMA's
 *   (profiles/r03_x6_vs_fp32_mfma_gemm_probe.txt; tests hold 2e-5 of the output scale). */
int vrnet_mlp_fused_ok(int C, int HID, long M);
long vrnet_mlp_pack_bytes(int C, int HID, int precision);
int vrnet_mlp_pack_f32(const float* w1, const float* w2, int C, int HID, int precision, void* pack_fwd, void* pack_bwd,
                       void* stream);
int vrnet_mlp_fwd_f32(const float* x, long ldx, const void* pack_fwd, const float* b1, const float* b2, const float* res,
                      long ldres, const float* res_scale, float* y, long ldy, float* upre, long ldu, double* stats, long M,
                      int C, int HID, int precision, void* stream);
int vrnet_mlp_bwd_f32(const float* dy, long lddy, const float* dy_scale, const void* pack_bwd, const float* upre, long ldu,
                      float* h, long ldh, float* du, long lddu, float* dx, long lddx, long M, int C, int HID, int precision,
                      void* stream);
/* The backward kernel WITHOUT a stored pre-activation (round 5, ABI 9): u = W1 x + b1 is recomputed chunk by chunk from the forward's
 * input rows x (the normalised block input: fp32, row stride ldx) against pack = vrnet_mlp_pack_rc_f32's per-chunk
 * [fc2^T | fc1^T | fc1] planes; the forward (vrnet_mlp_fwd_f32 with upre = NULL) then writes no hidden-sized tensor and the
 * backward reads one less.  precision 2: fp32 h / du, bit-identical to vrnet_mlp_bwd_f32 on the stored u; 4: bf16-rounded
 * operands, bf16 h / du.  Hidden widths up to 1024 (C = 64) / 1536 (C = 128): vrnet_mlp_rc_ok (ABI 10) is the predicate the
 * pack and the kernel check, for callers that decide in the FORWARD pass whether to store u. */
int vrnet_mlp_rc_ok(int C, int HID, long M);
long vrnet_mlp_pack_rc_bytes(int C, int HID, int precision);
int vrnet_mlp_pack_rc_f32(const float* w1, const float* w2, int C, int HID, int precision, void* pack, void* stream);
int vrnet_mlp_bwd_rc_f32(const float* dy, long lddy, const float* dy_scale, const void* pack, const float* x, long ldx,
                         const float* b1, float* h, long ldh, float* du, long lddu, float* dx, long lddx, long M, int C, int HID,
                         int precision, void* stream);

/* ---- per-(sample, channel) moments in fp64 ---------------------------------------------------------
 * out[b][c] = { sum_p x, sum_p x*x }                    (x2 == NULL)
 *           = { sum_p xm, sum_p xm*x2 }, xm = mask>0 ? x : 0   (x2 given; mask optional)
 * Feeds GroupNorm(1,C) (vr_coc.py:105-111), train-mode BatchNorm2d (normal_conv.py:45; vr_coc.py:310,329),
 * ShuffleAttention's avg-pool / per-channel GroupNorm (shuffle_attention.py:57,62), ECA's avg-pool (eca.py:17),
 * ASPP's global mean (coc_fpn_dual.py:91-92) and the layer-scale / bias / norm-parameter gradients. */
long vrnet_moments_workspace(int B, long HW, int C);
int vrnet_moments_f32(const float* x, long ldx, const float* x2, long ldx2, const float* mask, long ldm, int B,
                      long HW, int C, double* out, void* workspace, long workspace_bytes, void* stream);

/* out = pre(A*(x1 - S1) + D1) + E*(x2 - S2) + D2; pre: 0 none, 1 ReLU, 2 keep where masky > 0.  Coefficients are
 * indexed [b*coef_bstride + c] (0 = per channel, C = per sample and channel); NULL A/E = 1, NULL D1/D2/S1/S2 = 0,
 * NULL x1/x2 = term absent.  The shifts keep normalisation in torch's (x - mean) * scale order: the algebraically
 * equal A*x + (beta - mean*A) cancels catastrophically when |mean| >> std.  The apply step of GroupNorm / BatchNorm(+ReLU, + residual) forward AND backward,
 * ECA gating (eca.py:22), global-feature broadcast (coc_fpn_dual.py:96). */
int vrnet_affine_f32(const float* x1, long ld1, const float* A, const float* D1, const float* S1, int pre,
                     const float* masky, long ldm, const float* x2, long ld2, const float* E, const float* D2,
                     const float* S2, long coef_bstride, float* out, long ldo, int B, long HW, int C, int accumulate,
                     const float* add, long ldadd, void* stream);   /* add: out-of-place addend (out = ... + add) */

/* Coefficient kernels: moments -> affine coefficients, saved statistics, parameter gradients. */
/* GroupNorm(1,C), eps 1e-5 (vr_coc.py:105-111): y = A*(x - S) + D with A,D,S [B][C]; mean_rstd [B][2]. */
int vrnet_gn_coef_fwd(const double* mom, const float* gamma, const float* beta, float eps, int B, long HW, int C,
                      float* A, float* D, float* S, float* mean_rstd, void* stream);
/* The same coefficients from (sum, sum of squares) pairs, pairs_per_sample consecutive pairs per sample (the `stats`
 * output of vrnet_conv2d_f32).  gamma2 / beta2 (NULL = none): two-stream launch, samples [B/2, B) use them. */
int vrnet_gn_coef_from_pairs(const double* pairs, long pairs_per_sample, const float* gamma, const float* beta, float eps,
                             int B, long HW, int C, float* A, float* D, float* S, float* mean_rstd,
                             const float* gamma2, const float* beta2, void* stream);
/* vrnet_moments_f32 + vrnet_gn_coef_fwd in two launches (the per-sample totals come straight from the chunk partials);
 * workspace as vrnet_moments_workspace. */
int vrnet_gn_stats_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, int B, long HW, int C,
                       float* A, float* D, float* S, float* mean_rstd, const float* gamma2, const float* beta2,
                       void* workspace, long workspace_bytes, void* stream);
/* mom2 = moments(dy, x2 = x): dx = A*dy + E*(x - S) + D with A,E,D,S [B][C]; dgamma, dbeta [C].
 * gamma2 / dgamma2 / dbeta2 (NULL = none): two-stream launch, samples [B/2, B) belong to the second norm. */
int vrnet_gn_coef_bwd(const double* mom2, const float* mean_rstd, const float* gamma, int B, long HW, int C,
                      float* A, float* E, float* D, float* S, float* dgamma, float* dbeta, int accumulate,
                      const float* gamma2, float* dgamma2, float* dbeta2, void* stream);

/* GroupNorm(1, C) with the coefficient step INSIDE the apply kernels (vr_coc.py:105-111, 264-271: norm1 / norm2 of every
 * ClusterBlock sit on the critical chain of the step, so launches count).  C % 4 == 0, 16-byte rows.
 * vrnet_gn_apply_fwd: y = GN(x) in ONE launch from the (sum, sumsq) tile pairs the producing conv / fused Mlp left
 *   (`stats` of vrnet_conv2d_f32 / vrnet_mlp_fwd_f32: pairs_per_sample consecutive fp64 pairs per sample); every workgroup
 *   re-adds its sample's pairs in the same fixed order; mean_rstd [B][2] is written for the backward pass.
 * vrnet_gn_apply_bwd: TWO launches -- the moments kernel over (dy, dy * x), which also leaves gamma-weighted totals per
 *   chunk, then one kernel that finishes each sample's coefficients from those totals, writes out = dx (+ add) and, in an
 *   extra row of its grid, the parameter gradients dgamma / dbeta (+= when accumulate_params). */
int vrnet_gn_apply_fwd(const float* x, long ldx, const double* pairs, long pairs_per_sample, const float* gamma,
                       const float* beta, float eps, int B, long HW, int C, float* y, long ldy, float* mean_rstd, void* stream);
long vrnet_gn_bwd_workspace(int B, long HW, int C);
/* The apply step of vrnet_gn_apply_bwd alone (ONE launch) when the conv that produced dy already left its moments
 * (vrnet_conv_colstats with x2 = the GroupNorm input, gamma = the GroupNorm weight): partial [B*HW/32][C][2],
 * tile_totals [B*HW/32][ceil(C/32)][2]; HW % 32 == 0. */
int vrnet_gn_apply_bwd_from_partials(const float* dy, long lddy, const float* x, long ldx, const double* partial,
                                     const double* tile_totals, const float* mean_rstd, const float* gamma, int B, long HW,
                                     int C, const float* add, long ldadd, float* out, long ldo, float* dgamma, float* dbeta,
                                     int accumulate_params, void* stream);
/* Train-mode BatchNorm2d coefficients (y = A * (z - S) + D), running statistics and mean / rstd from the column partials
 * the producing conv left (vrnet_conv_colstats, x2 = NULL; partial [B*HW/32][C][2]; HW % 32 == 0): BaseConv's conv -> BN
 * (normal_conv.py:45-49) without a statistics pass over z. */
int vrnet_bn_coef_fwd_from_partials(const double* partial, const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, long long* num_batches_tracked, int B, long HW,
                                    int C, float* A, float* D, float* S, float* mean_rstd, void* stream);
int vrnet_gn_apply_bwd(const float* dy, long lddy, const float* x, long ldx, const float* mean_rstd, const float* gamma, int B,
                       long HW, int C, const float* add, long ldadd, float* out, long ldo, float* dgamma, float* dbeta,
                       int accumulate_params, void* workspace, long workspace_bytes, void* stream);
/* nn.BatchNorm2d: batch statistics + running-stat update (unbiased var, momentum) + num_batches_tracked += 1
 * when training, running statistics otherwise.  y = A*(z - S) + D with A,D,S [C]; mean_rstd [C][2]. */
int vrnet_bn_coef_fwd(const double* mom, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, long long* num_batches_tracked, int training, int B,
                      long HW, int C, float* A, float* D, float* S, float* mean_rstd, void* stream);
/* Train-mode BatchNorm in two launches instead of three: vrnet_moments_f32's first kernel over z, then ONE kernel that
 * adds the chunk partials per channel, updates the running statistics and writes the coefficients (no [B][C][2] table, no
 * separate reduce launch).  workspace as vrnet_moments_workspace.  The backward counterpart does the same for
 * moments(dy, x2 = z, mask). */
int vrnet_bn_stats_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, long long* num_batches_tracked, int B, long HW, int C,
                       float* A, float* D, float* S, float* mean_rstd, void* workspace, long workspace_bytes, void* stream);
int vrnet_bn_stats_bwd(const float* dy, long lddy, const float* z, long ldz, const float* mask, long ldm,
                       const float* mean_rstd, const float* gamma, int training, int B, long HW, int C, float* A, float* E,
                       float* D, float* S, float* dgamma, float* dbeta, int accumulate, void* workspace,
                       long workspace_bytes, void* stream);
/* BatchNorm backward of y = ReLU(BN(z)) WITHOUT reading y (round 4; BaseConv, normal_conv.py:36-49): the mask [y > 0] is
 * recomputed from z with the forward coefficients, fwd_A (z - fwd_S) + fwd_D evaluated as the single fused multiply-add
 * the forward apply used (same bits), which removes one of three tensor reads from the moments pass and from the apply
 * pass.  vrnet_bn_stats_bwd_zmask: as vrnet_bn_stats_bwd;  vrnet_bn_apply_bwd_zmask: dz = [mask] (A dy) + E (z - S) + D. */
int vrnet_bn_stats_bwd_zmask(const float* dy, long lddy, const float* z, long ldz, const float* fwd_A, const float* fwd_D,
                             const float* fwd_S, const float* mean_rstd, const float* gamma, int training, int B, long HW,
                             int C, float* A, float* E, float* D, float* S, float* dgamma, float* dbeta, int accumulate,
                             void* workspace, long workspace_bytes, void* stream);
int vrnet_bn_apply_bwd_zmask(const float* dy, long lddy, const float* z, long ldz, const float* fwd_A, const float* fwd_D,
                             const float* fwd_S, const float* A, const float* E, const float* D, const float* S, float* dz,
                             long lddz, int B, long HW, int C, void* stream);
/* mom2 = moments(dy, x2 = z, mask = relu output): dz = A*dy' + E*(z - S) + D with A,E,D,S [C]. */
int vrnet_bn_coef_bwd(const double* mom2, const float* mean_rstd, const float* gamma, int training, int B, long HW,
                      int C, float* A, float* E, float* D, float* S, float* dgamma, float* dbeta, int accumulate,
                      void* stream);
/* eca_block (backbone/attention_modules/eca.py:16-22): gate[b][c] = sigmoid(conv1d_k(mean_hw x)). */
int vrnet_eca_coef_fwd(const double* mom, const float* wk, int k, int B, long HW, int C, float* gate, void* stream);
/* mom2 = moments(dy, x2 = x): dx = gate*dy + F[b][c]; dwk [k].  F or dwk may be NULL (ABI 9): only the other half is computed
 * (dwk, the Conv1d weight gradient, is needed by nothing on the backward chain: a caller may issue it on a side stream). */
int vrnet_eca_coef_bwd(const double* mom2, const double* mom, const float* gate, const float* wk, int k, int B,
                       long HW, int C, float* F, float* dwk, int accumulate, void* stream);
/* Layer scale x + ls*o (vr_coc.py:266-271), mom2 = moments(dx, x2 = o): dls[c] = sum dx*o; dbias = ls * sum dx.
 * pair = 1: two-stream launch, samples [B/2, B) reduce into (dls2, dbias2) with ls2. */
int vrnet_ls_coef_bwd(const double* mom2, const float* ls, int B, int C, float* dls, float* dbias, int accumulate,
                      int pair, const float* ls2, float* dls2, float* dbias2, void* stream);
int vrnet_moments_to_float(const double* mom, float* out, long n, double scale, int which, void* stream);

/* ---- non-overlapping patch embedding as gather + GEMM ---------------------------------------------------
 * PointRecuder(patch_size=4, stride=4) over cat([x, fea_pos]) (backbone/fusion/vr_coc.py:83-102, 424-430, 583-586):
 * P[b,oy,ox,(ky*k+kx)*(C+CP)+c] = cat(x, pos)[b, oy*k+ky, ox*k+kx, c]  (pos: (H,W,CP) buffer shared by the batch,
 * NULL when CP == 0); the projection is then vrnet_conv2d_f32 with a 1x1 kernel over k*k*(C+CP) channels and the
 * weight in OHWI order (vrnet_weight_ohwi_f32 dir 0).  patch_scatter is the adjoint w.r.t. x;
 * vrnet_weight_ohwi_f32 dir 1 maps the OHWI weight gradient back to the parameter's OIHW layout. */
int vrnet_patch_gather_f32(const float* x, long ldx, const float* pos, float* out, int B, int H, int W, int C, int CP,
                           int k, void* stream);
int vrnet_patch_scatter_f32(const float* dp, float* dx, long lddx, int B, int H, int W, int C, int CP, int k,
                            int accumulate, void* stream);
int vrnet_weight_ohwi_f32(const float* src, float* dst, int Cout, int Cin, int kh, int kw, int dir, int accumulate,
                          void* stream);

/* ---- layout ---------------------------------------------------------------------------------------- */
/* dst[r*ldd + c*dcs] (+)= src[r*lds + c*scs]: torch.cat + shuffle_channels(groups=2) (vr_coc.py:70-80,
 * coc_fpn_dual.py:120-130) written straight into the consumer's buffer, and their adjoints. */
/* torch.cat([a, b], 1) (+ shuffle_channels(groups = 2) when interleave: channel 2 j = a_j, 2 j + 1 = b_j; vr_coc.py:70-80,
 * coc_fpn_dual.py:120-130) in ONE launch -- dir 0: cat (rows, Ca + Cb) = a | b -- and its adjoint -- dir 1: a (+)= its
 * channels of cat, b (+)= its channels (accumulate_a / accumulate_b; a or b NULL: that half is skipped). */
int vrnet_cat2_f32(float* a, long lda, int Ca, float* b, long ldb, int Cb, float* cat, long ldc, long rows, int interleave,
                   int dir, int accumulate_a, int accumulate_b, void* stream);
int vrnet_copy_channels_f32(const float* src, long lds, int scs, float* dst, long ldd, int dcs, long rows, int C,
                            int accumulate, void* stream);
int vrnet_nchw_to_nhwc_f32(const float* src, float* dst, long ldd, int B, int C, long HW, void* stream);
int vrnet_nhwc_to_nchw_f32(const float* src, long lds, float* dst, int B, int C, long HW, int accumulate, void* stream);
int vrnet_add_f32(float* dst, const float* src, long n, void* stream);
int vrnet_fill_f32(float* dst, float value, long n, void* stream);

/* ---- Context-Cluster core ---------------------------------------------------------------------------
 * Replaces Cluster.forward between fc1/fc_v and fc2 (vr_coc.py:158-190; pairwise_cos_sim :114-125) and its
 * autograd.  f, v, out: (B,H,W,E*D) NHWC; head e owns channels [e*D,(e+1)*D); regions are the fold x fold
 * tiles of the map (fold = 1: the whole map), D % 4 == 0, D <= 32.  Regions of <= 256 points (every backbone
 * stage at 512 px) stay in registers; larger ones (neck p3 at 512 px, everything at 1024 px) use a streaming kernel.
 * idx: (B,H,W,E) u8 hard assignment (first maximum, as torch.max(dim)); wgt: (B,H,W,E) similarity of the
 * assigned centre (optional for regions of <= 256 points, required above).  alpha, beta: device scalars (sim_alpha, sim_beta, :148-149).
 * alpha2 / beta2 (NULL = none): two-stream launch, samples [B/2, B) belong to a second Cluster module. */
int vrnet_cluster_fwd_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                          float* out, long ldo, unsigned char* idx, float* wgt, int B, int H, int W, int E, int D,
                          int fold, const float* alpha2, const float* beta2, void* stream);
/* the same forward with the assignment GIVEN (idx is read): parity work, where the numerically tied points of the arg-max
 * (vr_coc.py:173-176) must be decided the same way on both sides of a comparison */
int vrnet_cluster_fwd_forced_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                                 float* out, long ldo, const unsigned char* idx, float* wgt, int B, int H, int W, int E, int D,
                                 int fold, const float* alpha2, const float* beta2, void* stream);
long vrnet_cluster_bwd_workspace(int B, int E, int fold);                          /* regions of <= 256 points */
long vrnet_cluster_bwd_workspace2(int B, int H, int W, int E, int fold);            /* any region size */
/* Recomputes the forward from f, v with the saved assignment idx; df, dv share row stride lddf.
 * dalpha == dbeta == NULL (single-stream launches, also of vrnet_cluster_bwd_planes_f32; ABI 9): the per-workgroup (d alpha,
 * d beta) partials stay in the first 2 * B * E * fold^2 floats of `workspace` and no finishing launch runs -- the caller keeps
 * that workspace to itself and reduces a whole section's modules later with ONE vrnet_cluster_ab_reduce_multi. */
int vrnet_cluster_bwd_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                          const unsigned char* idx, const float* dout, long lddo, float* df, float* dv, long lddf,
                          float* dalpha, float* dbeta, int accumulate_ab, int B, int H, int W, int E, int D,
                          int fold, const float* alpha2, const float* beta2, float* dalpha2, float* dbeta2,
                          void* workspace, long workspace_bytes, void* stream);

/* (d alpha, d beta) of n Cluster modules (vr_coc.py:148-149: sim_alpha, sim_beta are shared by all regions and heads of a
 * module) from the partials their backward launches left (above): host arrays of n workspaces, their B * E * fold^2, the
 * gradient scalars (device) and accumulate flags; one launch per 32 modules. */
int vrnet_cluster_ab_reduce_multi(int n, const void* const* partial, const long* blocks, float* const* dalpha,
                                  float* const* dbeta, const int* accumulate, void* stream);

/* ---- depthwise 3x3, stride 1, pad 1 (DWConv.dconv, normal_conv.py:26-27; head towers decouplehead.py:23-34)
 * w: [C][3][3] (the OIHW tensor of a groups=C conv).  flip = 1 gives the input gradient. */
int vrnet_dwconv3x3_f32(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int C,
                        int flip, int accumulate, void* stream);
long vrnet_dwconv3x3_wgrad_workspace(int B, int H, int W, int C);
int vrnet_dwconv3x3_wgrad_f32(const float* x, long ldx, const float* dy, long lddy, float* dw, int B, int H, int W,
                              int C, int accumulate, void* workspace, long workspace_bytes, void* stream);

/* ---- nn.Upsample(scale, 'bilinear', align_corners=True) (coc_fpn_dual.py:21) and its adjoint (gather form).
 * out_nchw / dy_nchw: the high-resolution side is a contiguous (B,C,OH,OW) tensor (the seg logits). */
int vrnet_upsample_bilinear_f32(const float* x, long ldx, float* y, long ldy, int B, int H, int W, int C, int scale,
                                int out_nchw, void* stream);
/* The same gather from the PRE-normalisation map z of CoCUpsample's 1x1 BaseConv (coc_fpn_dual.py:15-26): every tap is
 * ReLU(A (z - S) + D) evaluated on the fly (A, D, S per channel: the BatchNorm coefficients), so the low-resolution activation
 * is never stored.  Bit-identical to vrnet_affine_f32 (pre = 1) + vrnet_upsample_bilinear_f32. */
int vrnet_bn_relu_upsample_bilinear_f32(const float* z, long ldz, const float* A, const float* D, const float* S, float* y,
                                        long ldy, int B, int H, int W, int C, int scale, int out_nchw, void* stream);
int vrnet_upsample_bilinear_bwd_f32(const float* dy, long lddy, int dy_nchw, float* dx, long lddx, int B, int H, int W,
                                    int C, int scale, int accumulate, void* stream);

/* ---- ImageEnhanceByRadar gain (vr_coc.py:312-316) with data_normal (:59-67): batch-global min/max of the
 * ReLU'd radar projection p (contiguous, n elements), out = (1 + (p-min)/(max-min)) * x, and the backward
 * including the gradient through min and max (ties share it evenly, as torch's min()/max() backward). */
long vrnet_reduce_workspace(void);
int vrnet_minmax_f32(const float* p, long n, float* mm /* [2] */, void* workspace, long workspace_bytes, void* stream);
int vrnet_enhance_mul_f32(const float* p, const float* x, const float* mm, float* out, long n, void* stream);
/* vrnet_minmax_f32 + vrnet_enhance_mul_f32 in two launches instead of three (the final min / max step runs inside the apply
 * kernel); mm receives (min, max) for vrnet_enhance_bwd_f32.  workspace: vrnet_reduce_workspace() bytes.  (ABI 9) */
int vrnet_enhance_fwd_f32(const float* p, const float* x, float* mm, float* out, long n, void* workspace, long workspace_bytes,
                          void* stream);

/* ---- fused passes of the fusion blocks (csrc/fusion.hip, ABI 9): an elementwise pass that also leaves the partial sums the NEXT
 * reduction needs (contiguous NHWC tensors of n elements, row stride == C, C % 4 == 0, C <= 1024, 16-byte aligned).
 * vrnet_fusion_chunks(n, C): workgroups = entries of the column partials (0: shape not supported): colpart [chunks][C][2] fp64;
 * vrnet_fusion_fold_chunks(n, C): entries of mmpart [.][2] fp32 and sums4 [.][4] fp64.
 *   vrnet_bn_relu_minmax_f32     p = ReLU(A (z - S) + D) + the (min, max) partials of p         (vr_coc.py:308 + :59-67)
 *   vrnet_enhance_stats_f32      t = (1 + data_normal(p)) x + column (sum, sumsq) of t; mm <- (min, max)      (:314-315)
 *   vrnet_bn_relu_res_stats_f32  s = ReLU(A (z - S) + D) + res + column (sum, sumsq) of s                      (:355-357)
 *   vrnet_bn_bwd_enhance_f32     dt = A g + E (t - S) + D (BatchNorm backward apply) + the four sums of the gain's backward
 *   vrnet_enhance_bwd_stats_f32  dx (+)=, dp of the gain + column (sum dp', sum dp' z), dp' = dp [fA (z - fS) + fD > 0]
 * vrnet_bn_coef_{fwd,bwd}_from_chunks: the BatchNorm coefficient steps (vrnet_bn_coef_fwd_from_partials / the second half of
 * vrnet_bn_stats_bwd) from such column partials; count = values per channel (B * HW). */
int vrnet_fusion_chunks(long n, int C);
int vrnet_fusion_fold_chunks(long n, int C);      /* entries of mmpart / sums4 (folded again by every workgroup of the next kernel) */
int vrnet_bn_relu_minmax_f32(const float* z, const float* A, const float* D, const float* S, float* p, long n, int C, float* mmpart,
                             void* stream);
int vrnet_bn_relu_res_stats_f32(const float* z, const float* A, const float* D, const float* S, const float* res, float* s, long n,
                                int C, double* colpart, void* stream);
int vrnet_enhance_stats_f32(const float* p, const float* x, const float* mmpart, int nmm, float* mm, float* t, long n, int C,
                            double* colpart, void* stream);
int vrnet_bn_bwd_enhance_f32(const float* g, const float* t, const float* A, const float* E, const float* D, const float* S,
                             const float* x, const float* p, const float* mm, float* dt, long n, int C, double* sums4, void* stream);
int vrnet_enhance_bwd_stats_f32(const float* dt, const float* x, const float* p, const float* mm, const double* sums4, int nsums,
                                const float* z, const float* fA, const float* fD, const float* fS, float* dx, float* dp, long n, int C,
                                int accumulate_dx, double* colpart, void* stream);
/* ds = A g + E (s - S) + D (backward apply of a BatchNorm without ReLU) + column (sum ds', sum ds' z), ds' = ds [fA (z - fS) + fD > 0]:
 * the moments of the BatchNorm + ReLU in front of it (vr_coc.py:355-357 backwards) */
int vrnet_bn_bwd_next_stats_f32(const float* g, const float* s, const float* A, const float* E, const float* D, const float* S,
                                const float* z, const float* fA, const float* fD, const float* fS, float* ds, long n, int C,
                                double* colpart, void* stream);
int vrnet_bn_coef_fwd_from_chunks(const double* partial, int nchunks, long count, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, long long* num_batches_tracked, int C,
                                  float* A, float* D, float* S, float* mean_rstd, void* stream);
int vrnet_bn_coef_bwd_from_chunks(const double* partial, int nchunks, long count, const float* mean_rstd, const float* gamma,
                                  int training, int C, float* A, float* E, float* D, float* S, float* dgamma, float* dbeta,
                                  int accumulate, void* stream);
int vrnet_enhance_bwd_f32(const float* dt, const float* x, const float* p, const float* mm, float* dx, float* dp,
                          long n, int accumulate_dx, void* workspace, long workspace_bytes, void* stream);

/* ---- ShuffleAttention (backbone/attention_modules/shuffle_attention.py:48-72) ---------------------------
 * mom = moments(x).  y[dst(q)] = x[q] * sigmoid(P[b][q]*(x[q] - Mn[b][q]) + Q[b][q]); dst() is the final 2-group channel
 * shuffle; the G-group split and the channel/spatial halves are index arithmetic on q.
 * Parameters: cweight, cbias, sweight, sbias, gn.weight, gn.bias, each [C/(2G)]. */
int vrnet_sa_coef_fwd(const double* mom, const float* cw, const float* cb, const float* sw, const float* sb,
                      const float* gnw, const float* gnb, int B, long HW, int C, int G, float* P, float* Q, float* Mn,
                      void* stream);
int vrnet_sa_apply_f32(const float* x, long ldx, const float* P, const float* Q, const float* Mn, float* y, long ldy,
                       int B, long HW, int C, void* stream);
/* cat = shuffle_channels(cat([ShuffleAttention(x), r], 1), 2) (vr_coc.py:343-349) and mom[b][c] = (sum of cat[b, :, c], 0) for the ECA
 * gate behind it (:350): vrnet_sa_apply_f32 + vrnet_cat2_f32 + the pass of vrnet_moments_f32 as one launch (+ the chunk reduce).
 * x, r: (B, HW, C); cat: (B, HW, 2 C); C % 4 == 0, C <= 512; workspace: vrnet_moments_workspace(B, HW, 2 C).  (ABI 9) */
int vrnet_sa_cat_sums_f32(const float* x, long ldx, const float* P, const float* Q, const float* Mn, const float* r, long ldr,
                          float* cat, long ldc, int B, long HW, int C, double* mom, void* workspace, long workspace_bytes,
                          void* stream);
long vrnet_sa_bwd_workspace(int B, long HW, int C);
int vrnet_sa_bwd_f32(const float* dy, long lddy, const float* x, long ldx, const float* P, const float* Q,
                     const float* Mn, const double* mom, const float* cw, const float* cb, const float* sw, const float* sb,
                     const float* gnw, const float* gnb, float* dx, long lddx, float* dcw, float* dcb, float* dsw,
                     float* dsb, float* dgnw, float* dgnb, float* EF /* [2][B][C] scratch */, int B, long HW, int C,
                     int G, int accumulate_dx, int accumulate_params, void* workspace, long workspace_bytes,
                     void* stream);

/* ---- training-step updates over ALL parameters in one launch ----------------------------------
 * Tensor table (device memory): addrs[k*n + t] = address of role k of tensor t, sizes[t] = element count; the work
 * list (chunk_tensor[c], chunk_index[c]) cuts every tensor into chunks of chunk_elems (multiple of 256) elements.
 * vrnet_mt_sgd_f32   roles {param, grad, momentum_buffer}: torch.optim.SGD(momentum, nesterov) as train.py:468-473
 *                    builds it; weight_decay[t] per tensor (group pg1 only, train.py:472); first_step = buffers unset.
 * vrnet_mt_adam_f32  roles {param, grad, exp_avg, exp_avg_sq}: torch.optim.Adam(betas=(momentum, 0.999)), step from 1.
 * vrnet_mt_ema_f32   roles {ema, model}: ModelEMA.update, nets/yolo_training.py:465-475 (v *= d; v += (1-d)*model). */
int vrnet_mt_sgd_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                     const float* weight_decay, int n_tensors, int n_chunks, int chunk_elems, float lr, float momentum,
                     int nesterov, int first_step, void* stream);
int vrnet_mt_adam_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                      const float* weight_decay, int n_tensors, int n_chunks, int chunk_elems, float lr, float beta1,
                      float beta2, float eps, int step, void* stream);
int vrnet_mt_ema_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                     int n_tensors, int n_chunks, int chunk_elems, float decay, void* stream);
/* roles {dst, src}: dst = src for every tensor of the table in one launch (the concatenated fc1 | fc_v weights of all
 * Cluster modules, vr_coc.py:145-147, so that both 1x1 convs of a block run as one GEMM). */
int vrnet_mt_copy_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                      int n_tensors, int n_chunks, int chunk_elems, void* stream);

/* ---- box decode ----------------------------------------------------------------------------
 * decode_outputs, utils/utils_bbox.py:32-84: levels[l] = raw head map (B, C = 5+num_classes, hs[l], ws[l]) NCHW (the
 * hot path's det outputs); out (B, sum_l hs*ws, C): [cx/in_w, cy/in_h, w/in_w, h/in_h, sigmoid(obj), sigmoid(cls)...],
 * anchors level-major then row-major, stride_l = input_h / hs[l].  `levels`, `hs`, `ws` are HOST arrays. */
int vrnet_decode_outputs_f32(const float* const* levels, const int* hs, const int* ws, int n_levels, int B, int C,
                             float input_h, float input_w, float* out, void* stream);

/* ---- input formats -----------------------------------------------------------------------------
 * What YoloDataset.__getitem__ / yolo_dataset_collate (utils/dataloader.py:88-107, 440-457) do to a letterboxed batch,
 * from BYTES: img (B, H, W, 3) u8 RGB -> images (B, 3, H, W) f32 = ((v / 255) - mean) / std evaluated in double and rounded
 * once (utils_seg/utils.py:43-47 preprocess_input on a float64 array, then FloatTensor: bit-identical); png (B, H, W) u8
 * labels -> png_out (B, H, W) int64 with labels >= num_classes_seg set to the ignore class num_classes_seg (:96-97) and
 * onehot (B, H, W, num_classes_seg + 1) f32 (:103-105).  img or png may be NULL (that half is skipped), as may png_out or
 * onehot. */
int vrnet_batch_formats_u8(const unsigned char* img, const unsigned char* png, int B, int H, int W, int num_classes_seg,
                           float* images, long long* png_out, float* onehot, void* stream);

/* ---- training losses on the path's outputs: value + gradient w.r.t. the head outputs ------------
 * vrnet_yolo_loss_f32: YOLOLoss (nets/yolo_training.py:60-427): decode (:99-111), SimOTA assignment per image
 *   (get_assignments :200-264, get_in_boxes_info :291-368, dynamic_k_matching :370-427), IoU / objectness / class
 *   losses (:113-198, IOUloss :13-57).  levels[l]: raw (B, C = 5+nc, hs[l], ws[l]) NCHW maps; grads[l]: same shapes,
 *   receives grad_scale * d loss / d levels[l] (grads or grads[l] NULL = value only); labels (B, max_gt, 5) =
 *   [cx, cy, w, h, class] in input pixels, counts[b] boxes valid in image b (0 allowed, :145-149).
 *   out[5] = {loss, num_fg, sum iou loss, sum obj loss, sum cls loss}; optional per-anchor assignment outputs
 *   fg_out (B,A) u8, matched_out (B,A) int (-1 = background), piou_out (B,A).  levels/grads/hs/ws/strides: HOST arrays.
 * vrnet_seg_loss_f32: CE_Loss (focal = 0), Focal_Loss (focal = 1) or no main term (focal = -1) (+ Dice_loss when dice = 1)
 *   (nets/deeplabv3_training.py:9-59; combination utils/utils_fit.py:96-103) on logits x (B, C, H, W) NCHW already at
 *   label size, png (B, H, W) int64 with ignore index C, onehot (B, H, W, C+1) float, weights[C] or NULL.
 *   out[3] = {main, dice, main + dice}; dx (NULL = value only) = grad_scale * d(main + dice)/dx. */
long vrnet_yolo_loss_workspace(int B, long n_anchors, int max_gt, int num_classes);
int vrnet_yolo_loss_f32(const float* const* levels, float* const* grads, const int* hs, const int* ws, const float* strides,
                        int n_levels, int B, int C, const float* labels, const int* counts, int max_gt, float grad_scale,
                        float* out, unsigned char* fg_out, int* matched_out, float* piou_out, void* workspace,
                        long workspace_bytes, void* stream);
long vrnet_seg_loss_workspace(int B, int C, long HW);
int vrnet_seg_loss_f32(const float* x, const long long* png, const float* onehot, const float* weights, int B, int C,
                       long HW, int focal, int dice, float alpha, float gamma, float beta, float smooth, float grad_scale,
                       float* out, float* dx, void* workspace, long workspace_bytes, void* stream);

/* ---- the synthetic backward driver: loss[0] = sum_k mean(t_k^2) over k <= 8 contiguous fp32 tensors -- the det maps
 * and the seg logits -- and its gradient grad_k = (2 g / n_k) t_k, g = upstream gradient of the scalar (device).  t, n, grad:
 * HOST arrays of device pointers / element counts.  Replaces the eager `sum((d * d).mean()) + (seg * seg).mean()` of the
 * reference's own smoke test (vr_coc.py:817-830 builds the same kind of scalar) and its autograd: ~27 launches -> 3. */
long vrnet_mean_square_workspace(int k, const long* n);
int vrnet_mean_square_f32(int k, const void* const* t, const long* n, float* loss, void* workspace, long workspace_bytes,
                          void* stream);
int vrnet_mean_square_bwd_f32(int k, const void* const* t, const long* n, const float* g, void* const* grad, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* VRNET_HIP_H */
