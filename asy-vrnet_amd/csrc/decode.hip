// Box decode of the detection head outputs (inference side of the hot path's det maps):
// utils/utils_bbox.py:32-84 `decode_outputs` -- flatten + concat the levels to (B, A, 5+nc), sigmoid on channels
// >= 4, xy = (xy + grid) * stride, wh = exp(wh) * stride, then x,w / input_w and y,h / input_h.
// The reference materialises the concat, the grids and the strides; here one thread owns one (image, anchor) and
// reads its 5+nc channels straight from the NCHW level maps (coalesced over anchors).  stride = input_h / h for BOTH
// axes, as the reference computes it (:65).
#include "common.h"

namespace {

constexpr int MAXL = 8;
struct DecodeArgs {
  const float* lvl[MAXL];
  int h[MAXL], w[MAXL], a0[MAXL + 1];   // a0: first anchor of each level, a0[nl] = total
  int nl, B, C;
  float in_h, in_w;
  float* out;
};

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs p) {
  const int A = p.a0[p.nl];
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)p.B * A) return;
  const int b = e / A, a = e - (long)b * A;
  int l = 0;
  while (l + 1 < p.nl && a >= p.a0[l + 1]) ++l;
  const int pix = a - p.a0[l], hw = p.h[l] * p.w[l];
  const int gy = pix / p.w[l], gx = pix - gy * p.w[l];
  const float stride = p.in_h / (float)p.h[l];
  const float* src = p.lvl[l] + (long)b * p.C * hw + pix;
  float* dst = p.out + e * p.C;
  for (int c = 0; c < p.C; ++c) {
    float v = src[(long)c * hw];
    if (c == 0) v = (v + (float)gx) * stride / p.in_w;
    else if (c == 1) v = (v + (float)gy) * stride / p.in_h;
    else if (c == 2) v = expf(v) * stride / p.in_w;
    else if (c == 3) v = expf(v) * stride / p.in_h;
    else v = 1.0f / (1.0f + expf(-v));
    dst[c] = v;
  }
}

}  // namespace

extern "C" int vrnet_decode_outputs_f32(const float* const* levels, const int* hs, const int* ws, int n_levels, int B,
                                        int C, float input_h, float input_w, float* out, void* stream) {
  VR_CHECK_ARG(levels && hs && ws && out && n_levels >= 1 && n_levels <= MAXL && B > 0 && C >= 5 && input_h > 0 &&
                   input_w > 0, "decode_outputs: bad arguments (1..%d levels, >= 5 channels)", MAXL);
  DecodeArgs p{};
  long A = 0;
  for (int l = 0; l < n_levels; ++l) {
    VR_CHECK_ARG(levels[l] && hs[l] > 0 && ws[l] > 0, "decode_outputs: bad level %d", l);
    p.lvl[l] = levels[l]; p.h[l] = hs[l]; p.w[l] = ws[l]; p.a0[l] = (int)A;
    A += (long)hs[l] * ws[l];
  }
  VR_CHECK_ARG(A * B < (1L << 31), "decode_outputs: too many anchors");
  p.a0[n_levels] = (int)A;
  p.nl = n_levels; p.B = B; p.C = C; p.in_h = input_h; p.in_w = input_w; p.out = out;
  hipLaunchKernelGGL(decode_kernel, dim3(vr_cdiv(A * B, 256)), dim3(256), 0, vr_stream(stream), p);
  VR_LAUNCH_CHECK("decode_outputs");
  return VR_OK;
}
