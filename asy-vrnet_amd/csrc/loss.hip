// Training losses on the hot path's outputs, forward value AND gradient w.r.t. the head outputs in one call, with no
// host synchronisation (the reference loops over images and ground-truth boxes in Python with .item() syncs,
// nets/yolo_training.py:143,392,413).
//   * vrnet_yolo_loss_f32: YOLOLoss.forward / get_losses / get_assignments / dynamic_k_matching (SimOTA),
//     nets/yolo_training.py:60-427, IOUloss :13-57.
//   * vrnet_seg_loss_f32: CE_Loss / Focal_Loss (+ Dice_loss), nets/deeplabv3_training.py:9-59.
// Every reduction has a fixed order (fp64 block partials, integer atomics only): results are reproducible.
#include "common.h"

#include <cfloat>

namespace {

// ============================================================================================== detection
constexpr int YMAXL = 8, YMAXC = 32;
struct YoloArgs {
  const float* lvl[YMAXL];
  float* grad[YMAXL];
  int h[YMAXL], w[YMAXL], a0[YMAXL + 1];
  float stride[YMAXL];
  int nl, B, C, NC, A, G;
  const float* labels;   // [B][G][5] cx, cy, w, h, cls (pixels)
  const int* counts;     // [B]
  float grad_scale;
  // workspace
  float* pred;           // [B][A][4]
  float* clsterm;        // [B][A][NC]   BCE(sqrt(sig(cls) sig(obj)), onehot(c)) summed over classes
  unsigned char* cand;   // [B][A]
  float* cost;           // [B][G][A]
  float* iou;            // [B][G][A]
  int* match;            // [B][G][10]
  int* kcount;           // [B][G]
  int* cnt;              // [B][A]
  int* owner;            // [B][A]
  int* numfg;            // [1]
  double* partial;       // [blocks][3]
  // outputs
  float* out;            // [5] loss, num_fg, iou, obj, cls
  unsigned char* fg_out; int* matched_out; float* piou_out;
};

__device__ __forceinline__ void anchor_of(const YoloArgs& p, int a, int& l, int& gx, int& gy, int& pix) {
  l = 0;
  while (l + 1 < p.nl && a >= p.a0[l + 1]) ++l;
  pix = a - p.a0[l];
  gy = pix / p.w[l];
  gx = pix - gy * p.w[l];
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// per (image, anchor): decoded box, class cost terms, candidate flag (in any box or any centre, :291-368); clears the
// per-anchor match bookkeeping.
__global__ __launch_bounds__(256) void yolo_prep_kernel(const YoloArgs p) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e == 0) *p.numfg = 0;
  if (e >= (long)p.B * p.A) return;
  const int b = e / p.A, a = e - (long)b * p.A;
  int l, gx, gy, pix;
  anchor_of(p, a, l, gx, gy, pix);
  const int hw = p.h[l] * p.w[l];
  const float s = p.stride[l];
  const float* src = p.lvl[l] + (long)b * p.C * hw + pix;
  const float bx = (src[0] + (float)gx) * s, by = (src[(long)hw] + (float)gy) * s;
  const float bw = expf(src[2L * hw]) * s, bh = expf(src[3L * hw]) * s;
  float* pr = p.pred + e * 4;
  pr[0] = bx; pr[1] = by; pr[2] = bw; pr[3] = bh;
  const float so = sigm(src[4L * hw]);
  float lp[YMAXC], l1[YMAXC], L0 = 0.f;
  for (int c = 0; c < p.NC; ++c) {
    const float q = sqrtf(sigm(src[(5L + c) * hw]) * so);
    lp[c] = fmaxf(logf(q), -100.f);             // F.binary_cross_entropy clamps its logs at -100
    l1[c] = fmaxf(logf(1.f - q), -100.f);
    L0 += l1[c];
  }
  for (int c = 0; c < p.NC; ++c) p.clsterm[e * p.NC + c] = -lp[c] - (L0 - l1[c]);
  const float xc = ((float)gx + 0.5f) * s, yc = ((float)gy + 0.5f) * s;
  bool any = false;
  const int ng = p.counts[b];
  for (int g = 0; g < ng; ++g) {
    const float* gt = p.labels + ((long)b * p.G + g) * 5;
    const float bl = xc - (gt[0] - 0.5f * gt[2]), br = (gt[0] + 0.5f * gt[2]) - xc;
    const float bt = yc - (gt[1] - 0.5f * gt[3]), bb = (gt[1] + 0.5f * gt[3]) - yc;
    const float cl = xc - (gt[0] - 2.5f * s), cr = (gt[0] + 2.5f * s) - xc;
    const float ct = yc - (gt[1] - 2.5f * s), cb = (gt[1] + 2.5f * s) - yc;
    any = any || fminf(fminf(bl, bt), fminf(br, bb)) > 0.f || fminf(fminf(cl, ct), fminf(cr, cb)) > 0.f;
  }
  p.cand[e] = any ? 1 : 0;
  p.cnt[e] = 0;
  p.owner[e] = 0x7fffffff;
}

struct VI { float v; int i; };
template <bool MAX>
__device__ __forceinline__ bool better(const VI& a, const VI& b) {
  return MAX ? (a.v > b.v || (a.v == b.v && a.i < b.i)) : (a.v < b.v || (a.v == b.v && a.i < b.i));
}
template <bool MAX>
__device__ __forceinline__ VI block_best(VI x, VI* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    VI y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    if (better<MAX>(y, x)) x = y;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = x;
  __syncthreads();
  VI r = red[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (better<MAX>(red[k], r)) r = red[k];
  return r;
}

// one workgroup per (ground-truth box, image): pairwise IoU + cost against every candidate anchor (:214-258), then
// dynamic k = clamp(int(sum of the 10 largest IoUs), 1) and the k cheapest anchors (:381-394).  Ties go to the lower
// anchor index (torch.topk leaves tie order unspecified).
__global__ __launch_bounds__(256) void yolo_match_kernel(const YoloArgs p) {
  __shared__ unsigned taken[1024];     // one bit per anchor (A <= 32768)
  __shared__ VI red[4];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (g >= p.counts[b]) return;
  const float* gt = p.labels + ((long)b * p.G + g) * 5;
  const float gx = gt[0], gy = gt[1], gw = gt[2], gh = gt[3];
  const int gc = (int)gt[4];
  float* cost = p.cost + ((long)b * p.G + g) * p.A;
  float* iou = p.iou + ((long)b * p.G + g) * p.A;
  for (int a = tid; a < p.A; a += 256) {
    const long e = (long)b * p.A + a;
    float cv = INFINITY, iv = -1.f;
    if (p.cand[e]) {
      int l, ax, ay, pix;
      anchor_of(p, a, l, ax, ay, pix);
      const float s = p.stride[l];
      const float xc = ((float)ax + 0.5f) * s, yc = ((float)ay + 0.5f) * s;
      const float bl = xc - (gx - 0.5f * gw), br = (gx + 0.5f * gw) - xc, bt = yc - (gy - 0.5f * gh), bb = (gy + 0.5f * gh) - yc;
      const float cl = xc - (gx - 2.5f * s), cr = (gx + 2.5f * s) - xc, ct = yc - (gy - 2.5f * s), cb = (gy + 2.5f * s) - yc;
      const bool both = fminf(fminf(bl, bt), fminf(br, bb)) > 0.f && fminf(fminf(cl, ct), fminf(cr, cb)) > 0.f;
      const float* pr = p.pred + e * 4;
      const float tlx = fmaxf(gx - gw / 2, pr[0] - pr[2] / 2), tly = fmaxf(gy - gh / 2, pr[1] - pr[3] / 2);
      const float brx = fminf(gx + gw / 2, pr[0] + pr[2] / 2), bry = fminf(gy + gh / 2, pr[1] + pr[3] / 2);
      const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
      const float ai = (brx - tlx) * (bry - tly) * en;
      iv = ai / (gw * gh + pr[2] * pr[3] - ai);
      cv = p.clsterm[e * p.NC + gc] + 3.0f * (-logf(iv + 1e-8f)) + (both ? 0.f : 100000.0f);
    }
    cost[a] = cv;
    iou[a] = iv;
  }
  for (int i = tid; i < 1024; i += 256) taken[i] = 0u;
  __syncthreads();
  // ---- dynamic k
  float sum = 0.f;
  for (int it = 0; it < 10; ++it) {
    VI best{-INFINITY, 0x7fffffff};
    for (int a = tid; a < p.A; a += 256)
      if (!((taken[a >> 5] >> (a & 31)) & 1u)) {
        const VI c{iou[a], a};
        if (c.v >= 0.f && better<true>(c, best)) best = c;
      }
    best = block_best<true>(best, red);
    if (best.i == 0x7fffffff) break;                 // fewer than 10 candidates
    sum += best.v;
    if (tid == 0) taken[best.i >> 5] |= 1u << (best.i & 31);
    __syncthreads();
  }
  int k = (int)sum;
  if (k < 1) k = 1;
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) taken[i] = 0u;
  __syncthreads();
  int found = 0;
  for (int it = 0; it < k; ++it) {
    VI best{INFINITY, 0x7fffffff};
    for (int a = tid; a < p.A; a += 256)
      if (!((taken[a >> 5] >> (a & 31)) & 1u)) {
        const VI c{cost[a], a};
        if (c.v < INFINITY && better<false>(c, best)) best = c;
      }
    best = block_best<false>(best, red);
    if (best.i == 0x7fffffff) break;
    if (tid == 0) {
      taken[best.i >> 5] |= 1u << (best.i & 31);
      p.match[((long)b * p.G + g) * 10 + found] = best.i;
    }
    ++found;
    __syncthreads();
  }
  if (tid == 0) p.kcount[(long)b * p.G + g] = found;
}

// per match entry: count the matches of its anchor, remember the (lowest) matching box
__global__ void yolo_vote_kernel(const YoloArgs p) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)p.B * p.G * 10) return;
  const int j = e % 10;
  const long bg = e / 10;
  const int g = bg % p.G, b = bg / p.G;
  if (g >= p.counts[b] || j >= p.kcount[bg]) return;
  const long ea = (long)b * p.A + p.match[e];
  atomicAdd(&p.cnt[ea], 1);
  atomicMin(&p.owner[ea], g);
}

// per (image, anchor): an anchor claimed by several boxes goes to the box of least cost over ALL boxes (:400-407)
__global__ __launch_bounds__(256) void yolo_resolve_kernel(const YoloArgs p) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)p.B * p.A) return;
  const int b = e / p.A, a = e - (long)b * p.A;
  const int c = p.cnt[e];
  int g = -1;
  if (c == 1) g = p.owner[e];
  else if (c > 1) {
    float best = INFINITY;
    const int ng = p.counts[b];
    for (int q = 0; q < ng; ++q) {
      const float v = p.cost[((long)b * p.G + q) * p.A + a];
      if (v < best) { best = v; g = q; }
    }
  }
  p.owner[e] = g;
  if (g >= 0) atomicAdd(p.numfg, 1);
  if (p.fg_out) p.fg_out[e] = g >= 0;
  if (p.matched_out) p.matched_out[e] = g;
  if (p.piou_out) p.piou_out[e] = g >= 0 ? p.iou[((long)b * p.G + g) * p.A + a] : 0.f;
}

__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

// per (image, anchor): loss terms (:176-181) and their gradient w.r.t. the raw head outputs
__global__ __launch_bounds__(256) void yolo_loss_kernel(const YoloArgs p) {
  __shared__ double red[4][3];
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  double l_iou = 0, l_obj = 0, l_cls = 0;
  if (e < (long)p.B * p.A) {
    const int b = e / p.A, a = e - (long)b * p.A;
    int l, ax, ay, pix;
    anchor_of(p, a, l, ax, ay, pix);
    const int hw = p.h[l] * p.w[l];
    const float s = p.stride[l];
    const float* src = p.lvl[l] + (long)b * p.C * hw + pix;
    float* dst = p.grad[l] ? p.grad[l] + (long)b * p.C * hw + pix : nullptr;
    const int nfg = *p.numfg;
    const float gs = p.grad_scale / (float)(nfg > 1 ? nfg : 1);
    const int g = p.owner[e];
    const float t_obj = g >= 0 ? 1.f : 0.f;
    const float xo = src[4L * hw];
    l_obj = bce_logits(xo, t_obj);
    if (dst) dst[4L * hw] = gs * 2.f * (sigm(xo) - t_obj);
    if (g >= 0) {
      const float* gt = p.labels + ((long)b * p.G + g) * 5;
      const float piou = p.iou[((long)b * p.G + g) * p.A + a];
      const int gc = (int)gt[4];
      for (int c = 0; c < p.NC; ++c) {
        const float xc = src[(5L + c) * hw], t = c == gc ? piou : 0.f;
        l_cls += bce_logits(xc, t);
        if (dst) dst[(5L + c) * hw] = gs * 2.f * (sigm(xc) - t);
      }
      const float* pr = p.pred + e * 4;
      const float px = pr[0], py = pr[1], pw = pr[2], ph = pr[3];
      const float gx = gt[0], gy = gt[1], gw = gt[2], gh = gt[3];
      const float ptlx = px - pw / 2, ptly = py - ph / 2, pbrx = px + pw / 2, pbry = py + ph / 2;
      const float gtlx = gx - gw / 2, gtly = gy - gh / 2, gbrx = gx + gw / 2, gbry = gy + gh / 2;
      const float tlx = fmaxf(ptlx, gtlx), tly = fmaxf(ptly, gtly), brx = fminf(pbrx, gbrx), bry = fminf(pbry, gbry);
      const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
      const float I = (brx - tlx) * (bry - tly) * en;
      const float ap = pw * ph;
      const float U = ap + gw * gh - I + 1e-16f;
      const float iou = I / U;
      l_iou = 1.f - iou * iou;
      if (dst) {
        const float dL = -2.f * iou;                          // d loss / d iou
        const float dI = dL * (1.f / U + I / (U * U)), dap = dL * (-I / (U * U));
        // torch.max / torch.min send the gradient to the winning operand, half to each on a tie
        const float s_tlx = ptlx > gtlx ? 1.f : (ptlx == gtlx ? 0.5f : 0.f), s_tly = ptly > gtly ? 1.f : (ptly == gtly ? 0.5f : 0.f);
        const float s_brx = pbrx < gbrx ? 1.f : (pbrx == gbrx ? 0.5f : 0.f), s_bry = pbry < gbry ? 1.f : (pbry == gbry ? 0.5f : 0.f);
        const float d_tlx = -dI * (bry - tly) * en * s_tlx, d_brx = dI * (bry - tly) * en * s_brx;
        const float d_tly = -dI * (brx - tlx) * en * s_tly, d_bry = dI * (brx - tlx) * en * s_bry;
        const float dpx = d_tlx + d_brx, dpy = d_tly + d_bry;
        const float dpw = 0.5f * (d_brx - d_tlx) + dap * ph, dph = 0.5f * (d_bry - d_tly) + dap * pw;
        dst[0] = gs * dpx * s;
        dst[(long)hw] = gs * dpy * s;
        dst[2L * hw] = gs * dpw * pw;
        dst[3L * hw] = gs * dph * ph;
      }
    } else if (dst) {
      for (int c = 0; c < 4; ++c) dst[(long)c * hw] = 0.f;
      for (int c = 0; c < p.NC; ++c) dst[(5L + c) * hw] = 0.f;
    }
  }
  l_iou = wave_sum(l_iou); l_obj = wave_sum(l_obj); l_cls = wave_sum(l_cls);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = l_iou; red[wave][1] = l_obj; red[wave][2] = l_cls; }
  __syncthreads();
  if (threadIdx.x < 3)
    p.partial[(long)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void yolo_final_kernel(const YoloArgs p, int nblocks) {
  __shared__ double tot[3];
  if (threadIdx.x < 3) {
    double s = 0;
    for (int k = 0; k < nblocks; ++k) s += p.partial[(long)k * 3 + threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nfg = *p.numfg;
    const double den = nfg > 1 ? nfg : 1;
    p.out[0] = (float)((tot[0] + 2.0 * tot[1] + 2.0 * tot[2]) / den);
    p.out[1] = (float)nfg;
    p.out[2] = (float)tot[0]; p.out[3] = (float)tot[1]; p.out[4] = (float)tot[2];
  }
}

struct YoloPlan { long off[12]; long total; };
YoloPlan yolo_plan(int B, long A, int G, int NC) {
  YoloPlan pl{};
  long o = 0;
  auto take = [&](int i, long bytes) { pl.off[i] = o; o += (bytes + 255) / 256 * 256; };
  take(0, B * A * 4 * 4);                    // pred
  take(1, B * A * NC * 4);                   // clsterm
  take(2, B * A);                            // cand
  take(3, (long)B * G * A * 4);              // cost
  take(4, (long)B * G * A * 4);              // iou
  take(5, (long)B * G * 10 * 4);             // match
  take(6, (long)B * G * 4);                  // kcount
  take(7, B * A * 4);                        // cnt
  take(8, B * A * 4);                        // owner
  take(9, 256);                              // numfg
  take(10, vr_cdiv(B * A, 256) * 3 * 8);     // partial
  pl.total = o;
  return pl;
}

// ============================================================================================== segmentation
constexpr int SMAXC = 32;
struct SegArgs {
  const float* x; const long long* png; const float* onehot; const float* weights;
  int B, C, ignore; long HW;
  int focal, dice;
  float alpha, gamma, beta, smooth, grad_scale;
  double* partial;     // [blocks][3 + 3C]: ce_num, ce_den, focal_sum, tp[C], sp[C], st[C]
  double* coef;        // [3 + 2C]: 1/ce_den, 1/N, (unused), dtp[C], dsp[C]
  float* out;          // [3]: main, dice, total
  float* dx;
  int nblocks;
};

__device__ __forceinline__ void seg_softmax(const SegArgs& p, long b, long pix, float (&v)[SMAXC], float& lse) {
  const float* src = p.x + (b * p.C) * p.HW + pix;
  float m = -INFINITY;
  for (int c = 0; c < p.C; ++c) { v[c] = src[(long)c * p.HW]; m = fmaxf(m, v[c]); }
  float s = 0.f;
  for (int c = 0; c < p.C; ++c) s += expf(v[c] - m);
  lse = m + logf(s);
}

__global__ __launch_bounds__(256) void seg_reduce_kernel(const SegArgs p) {
  extern __shared__ double sred[];   // [4][K]
  const int K = 3 + 3 * p.C;
  double acc[3 + 3 * SMAXC];
  for (int i = 0; i < K; ++i) acc[i] = 0.0;
  const long total = (long)p.B * p.HW;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / p.HW, pix = e - b * p.HW;
    float v[SMAXC], lse;
    seg_softmax(p, b, pix, v, lse);
    const int t = p.focal >= 0 ? (int)p.png[e] : p.ignore;       // focal < 0: no main term (Dice alone)
    if (t != p.ignore) {
      const float w = p.weights ? p.weights[t] : 1.f;
      const float ce = lse - v[t];
      acc[0] += (double)(w * ce);
      acc[1] += (double)w;
      const float u = -(w * ce), pt = expf(u);
      acc[2] += (double)(-powf(1.f - pt, p.gamma) * (p.alpha * u));
    }
    if (p.dice) {
      const float* oh = p.onehot + e * (p.C + 1);
      for (int c = 0; c < p.C; ++c) {
        const float pc = expf(v[c] - lse), tc = oh[c];
        acc[3 + c] += (double)(tc * pc);
        acc[3 + p.C + c] += (double)pc;
        acc[3 + 2 * p.C + c] += (double)tc;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = 0; i < K; ++i) {
    const double s = wave_sum(acc[i]);
    if (lane == 0) sred[wave * K + i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += 256)
    p.partial[(long)blockIdx.x * K + i] = sred[i] + sred[K + i] + sred[2 * K + i] + sred[3 * K + i];
}

__global__ void seg_final_kernel(const SegArgs p) {
  extern __shared__ double tot[];
  const int K = 3 + 3 * p.C;
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    double s = 0;
    for (int k = 0; k < p.nblocks; ++k) s += p.partial[(long)k * K + i];
    tot[i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double N = (double)p.B * (double)p.HW;
    const double main_loss = p.focal < 0 ? 0.0 : (p.focal ? tot[2] / N : tot[0] / tot[1]);
    p.coef[0] = 1.0 / tot[1];
    p.coef[1] = 1.0 / N;
    double dice = 0.0;
    if (p.dice) {
      const double b2 = (double)p.beta * p.beta, sm = p.smooth;
      double mean = 0.0;
      for (int c = 0; c < p.C; ++c) {
        const double tp = tot[3 + c], sp = tot[3 + p.C + c], st = tot[3 + 2 * p.C + c];
        const double D = b2 * st + sp + sm, Nn = (1.0 + b2) * tp + sm;     // fn + fp + tp terms: tp cancels in D
        mean += Nn / D;
        p.coef[3 + c] = -(1.0 + b2) / (D * p.C);          // d dice / d tp_c
        p.coef[3 + p.C + c] = Nn / (D * D * p.C);         // d dice / d sp_c
      }
      dice = 1.0 - mean / p.C;
    }
    p.out[0] = (float)main_loss;
    p.out[1] = (float)dice;
    p.out[2] = (float)(main_loss + dice);
  }
}

__global__ __launch_bounds__(256) void seg_grad_kernel(const SegArgs p) {
  const long total = (long)p.B * p.HW;
  const float inv_den = (float)p.coef[0], inv_n = (float)p.coef[1];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long b = e / p.HW, pix = e - b * p.HW;
    float v[SMAXC], lse;
    seg_softmax(p, b, pix, v, lse);
    const int t = p.focal >= 0 ? (int)p.png[e] : p.ignore;
    float k_ce = 0.f;              // d main / d ce_i, with ce_i = lse - x_t
    if (t != p.ignore) {
      const float w = p.weights ? p.weights[t] : 1.f;
      if (p.focal == 0) k_ce = w * inv_den;
      else {
        const float u = -(w * (lse - v[t])), pt = expf(u), om = 1.f - pt;
        // L = -alpha (1-pt)^g u ;  dL/du = -alpha (1-pt)^(g-1) [(1-pt) - g u pt] ;  du/dce = -w
        const float dLdu = -p.alpha * powf(om, p.gamma - 1.f) * (om - p.gamma * u * pt);
        k_ce = dLdu * (-w) * inv_n;
      }
    }
    float a[SMAXC], dot = 0.f;
    if (p.dice) {
      const float* oh = p.onehot + e * (p.C + 1);
      for (int c = 0; c < p.C; ++c) {
        a[c] = oh[c] * (float)p.coef[3 + c] + (float)p.coef[3 + p.C + c];
        dot += a[c] * expf(v[c] - lse);
      }
    }
    float* dst = p.dx + (b * p.C) * p.HW + pix;
    for (int c = 0; c < p.C; ++c) {
      const float pc = expf(v[c] - lse);
      float g = k_ce * (pc - (c == t ? 1.f : 0.f));
      if (p.dice) g += pc * (a[c] - dot);
      dst[(long)c * p.HW] = p.grad_scale * g;
    }
  }
}

long seg_blocks(long n) {
  long b = vr_cdiv(n, 1024);
  return b > 2048 ? 2048 : (b < 1 ? 1 : b);
}

}  // namespace

extern "C" long vrnet_yolo_loss_workspace(int B, long n_anchors, int max_gt, int num_classes) {
  return yolo_plan(B, n_anchors, max_gt < 1 ? 1 : max_gt, num_classes).total;
}

extern "C" int vrnet_yolo_loss_f32(const float* const* levels, float* const* grads, const int* hs, const int* ws,
                                   const float* strides, int n_levels, int B, int C, const float* labels,
                                   const int* counts, int max_gt, float grad_scale, float* out, unsigned char* fg_out,
                                   int* matched_out, float* piou_out, void* workspace, long workspace_bytes,
                                   void* stream) {
  VR_CHECK_ARG(levels && hs && ws && strides && out && counts && workspace && n_levels >= 1 && n_levels <= YMAXL && B > 0 &&
                   C > 5 && C - 5 <= YMAXC && max_gt >= 0 && (max_gt == 0 || labels),
               "yolo_loss: bad arguments (1..%d levels, 1..%d classes)", YMAXL, YMAXC);
  YoloArgs p{};
  long A = 0;
  for (int l = 0; l < n_levels; ++l) {
    VR_CHECK_ARG(levels[l] && hs[l] > 0 && ws[l] > 0 && strides[l] > 0, "yolo_loss: bad level %d", l);
    p.lvl[l] = levels[l]; p.grad[l] = grads ? grads[l] : nullptr;
    p.h[l] = hs[l]; p.w[l] = ws[l]; p.stride[l] = strides[l]; p.a0[l] = (int)A;
    A += (long)hs[l] * ws[l];
  }
  VR_CHECK_ARG(A <= 32768, "yolo_loss: %ld anchors per image (limit 32768)", A);
  p.a0[n_levels] = (int)A;
  const int G = max_gt < 1 ? 1 : max_gt;
  p.nl = n_levels; p.B = B; p.C = C; p.NC = C - 5; p.A = (int)A; p.G = G;
  p.labels = labels; p.counts = counts; p.grad_scale = grad_scale;
  const YoloPlan pl = yolo_plan(B, A, G, p.NC);
  if (workspace_bytes < pl.total) {
    vr_set_error("yolo_loss: workspace %ld < %ld bytes", workspace_bytes, pl.total);
    return VR_ERR_WORKSPACE;
  }
  char* ws8 = reinterpret_cast<char*>(workspace);
  p.pred = reinterpret_cast<float*>(ws8 + pl.off[0]);
  p.clsterm = reinterpret_cast<float*>(ws8 + pl.off[1]);
  p.cand = reinterpret_cast<unsigned char*>(ws8 + pl.off[2]);
  p.cost = reinterpret_cast<float*>(ws8 + pl.off[3]);
  p.iou = reinterpret_cast<float*>(ws8 + pl.off[4]);
  p.match = reinterpret_cast<int*>(ws8 + pl.off[5]);
  p.kcount = reinterpret_cast<int*>(ws8 + pl.off[6]);
  p.cnt = reinterpret_cast<int*>(ws8 + pl.off[7]);
  p.owner = reinterpret_cast<int*>(ws8 + pl.off[8]);
  p.numfg = reinterpret_cast<int*>(ws8 + pl.off[9]);
  p.partial = reinterpret_cast<double*>(ws8 + pl.off[10]);
  p.out = out; p.fg_out = fg_out; p.matched_out = matched_out; p.piou_out = piou_out;
  hipStream_t st = vr_stream(stream);
  const int nb = (int)vr_cdiv((long)B * A, 256);
  hipLaunchKernelGGL(yolo_prep_kernel, dim3(nb), dim3(256), 0, st, p);
  if (max_gt > 0) {
    hipLaunchKernelGGL(yolo_match_kernel, dim3(G, B), dim3(256), 0, st, p);
    hipLaunchKernelGGL(yolo_vote_kernel, dim3(vr_cdiv((long)B * G * 10, 256)), dim3(256), 0, st, p);
  }
  hipLaunchKernelGGL(yolo_resolve_kernel, dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(yolo_loss_kernel, dim3(nb), dim3(256), 0, st, p);
  hipLaunchKernelGGL(yolo_final_kernel, dim3(1), dim3(64), 0, st, p, nb);
  VR_LAUNCH_CHECK("yolo_loss");
  return VR_OK;
}

extern "C" long vrnet_seg_loss_workspace(int B, int C, long HW) {
  return (seg_blocks((long)B * HW) * (3 + 3 * C) + 3 + 2 * C) * 8 + 512;
}

extern "C" int vrnet_seg_loss_f32(const float* x, const long long* png, const float* onehot, const float* weights, int B,
                                  int C, long HW, int focal, int dice, float alpha, float gamma, float beta, float smooth,
                                  float grad_scale, float* out, float* dx, void* workspace, long workspace_bytes,
                                  void* stream) {
  VR_CHECK_ARG(x && (png || focal < 0) && out && workspace && B > 0 && C > 0 && C <= SMAXC && HW > 0 && (!dice || onehot) &&
                   (focal >= 0 || dice),
               "seg_loss: bad arguments (<= %d classes; dice needs the one-hot labels; focal < 0 needs dice)", SMAXC);
  if (workspace_bytes < vrnet_seg_loss_workspace(B, C, HW)) {
    vr_set_error("seg_loss: workspace too small");
    return VR_ERR_WORKSPACE;
  }
  SegArgs p{};
  p.x = x; p.png = png; p.onehot = onehot; p.weights = weights; p.B = B; p.C = C; p.ignore = C; p.HW = HW;
  p.focal = focal; p.dice = dice; p.alpha = alpha; p.gamma = gamma; p.beta = beta; p.smooth = smooth;
  p.grad_scale = grad_scale; p.out = out; p.dx = dx;
  p.nblocks = (int)seg_blocks((long)B * HW);
  const int K = 3 + 3 * C;
  p.partial = reinterpret_cast<double*>(workspace);
  p.coef = p.partial + (long)p.nblocks * K;
  hipStream_t st = vr_stream(stream);
  hipLaunchKernelGGL(seg_reduce_kernel, dim3(p.nblocks), dim3(256), 4 * K * sizeof(double), st, p);
  hipLaunchKernelGGL(seg_final_kernel, dim3(1), dim3(128), K * sizeof(double), st, p);
  if (dx) hipLaunchKernelGGL(seg_grad_kernel, dim3(p.nblocks), dim3(256), 0, st, p);
  VR_LAUNCH_CHECK("seg_loss");
  return VR_OK;
}

// ---- the synthetic backward driver of the benchmark: L = sum_k mean(t_k^2) over up to MS_MAX tensors ----------
// (the det maps and the seg logits).  As eager torch this is ~27 elementwise / reduce launches between the forward and the
// backward pass, every one of them exposed; here: one pass for the value (fp64 partials, fixed order), one for the gradient
// dL/dt_k = (2 g / n_k) t_k with the upstream scalar g read on the device.
namespace {
constexpr int MS_MAX = 8;
struct MsTable {
  const float* t[MS_MAX];
  float* grad[MS_MAX];
  long n[MS_MAX];
  long first[MS_MAX + 1];      // first workgroup of each tensor
  int k;
};
constexpr int MS_PER_WG = 256 * 16;

__global__ __launch_bounds__(256) void mean_square_partial_kernel(const MsTable tb, double* partial) {
  __shared__ double red[4];
  int e = 0;
  while (e + 1 < tb.k && (long)blockIdx.x >= tb.first[e + 1]) ++e;
  const long base = ((long)blockIdx.x - tb.first[e]) * MS_PER_WG;
  const float* t = tb.t[e];
  const long n = tb.n[e];
  double s = 0.0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long idx = base + (long)i * 256 + threadIdx.x;
    if (idx < n) { const double v = (double)t[idx]; s += v * v; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)n;
}
__global__ __launch_bounds__(256) void mean_square_final_kernel(const double* partial, long nwg, float* loss) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < nwg; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void mean_square_grad_kernel(const MsTable tb, const float* g) {
  int e = 0;
  while (e + 1 < tb.k && (long)blockIdx.x >= tb.first[e + 1]) ++e;
  const long base = ((long)blockIdx.x - tb.first[e]) * MS_PER_WG;
  const float* t = tb.t[e];
  float* d = tb.grad[e];
  const long n = tb.n[e];
  const float sc = (float)(2.0 * (double)g[0] / (double)n);
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long idx = base + (long)i * 256 + threadIdx.x;
    if (idx < n) d[idx] = sc * t[idx];
  }
}
int ms_table(int k, const void* const* t, void* const* grad, const long* n, MsTable* tb, long* nwg) {
  long first = 0;
  tb->k = k;
  for (int i = 0; i < k; ++i) {
    tb->t[i] = reinterpret_cast<const float*>(t[i]);
    tb->grad[i] = grad ? reinterpret_cast<float*>(grad[i]) : nullptr;
    tb->n[i] = n[i];
    tb->first[i] = first;
    first += (n[i] + MS_PER_WG - 1) / MS_PER_WG;
  }
  tb->first[k] = first;
  *nwg = first;
  return 0;
}
}  // namespace

extern "C" long vrnet_mean_square_workspace(int k, const long* n) {
  long wg = 0;
  for (int i = 0; i < k; ++i) wg += (n[i] + MS_PER_WG - 1) / MS_PER_WG;
  return wg * 8 + 256;
}
/* loss[0] = sum_k mean(t_k^2) over k <= 8 contiguous fp32 tensors of n_k elements (host arrays of device pointers / sizes):
 * the fixed synthetic scalar that drives the backward pass of the benchmark and of the parity tests (the reference's
 * real losses are vrnet_yolo_loss_f32 / vrnet_seg_loss_f32). */
extern "C" int vrnet_mean_square_f32(int k, const void* const* t, const long* n, float* loss, void* workspace, long workspace_bytes,
                                     void* stream) {
  VR_CHECK_ARG(k > 0 && k <= MS_MAX && t && n && loss && workspace, "mean_square: bad arguments (<= %d tensors)", MS_MAX);
  for (int i = 0; i < k; ++i) VR_CHECK_ARG(t[i] && n[i] > 0, "mean_square: bad tensor %d", i);
  VR_CHECK_ARG(workspace_bytes >= vrnet_mean_square_workspace(k, n), "mean_square: workspace too small");
  MsTable tb{};
  long nwg;
  ms_table(k, t, nullptr, n, &tb, &nwg);
  VR_CHECK_ARG(nwg < (1L << 31), "mean_square: too large");
  hipStream_t st = vr_stream(stream);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(mean_square_partial_kernel, dim3((unsigned)nwg), dim3(256), 0, st, tb, partial);
  hipLaunchKernelGGL(mean_square_final_kernel, dim3(1), dim3(256), 0, st, partial, nwg, loss);
  VR_LAUNCH_CHECK("mean_square");
  return VR_OK;
}
/* grad_k = (2 g / n_k) t_k with the upstream gradient g of the scalar read from device memory: one launch. */
extern "C" int vrnet_mean_square_bwd_f32(int k, const void* const* t, const long* n, const float* g, void* const* grad, void* stream) {
  VR_CHECK_ARG(k > 0 && k <= MS_MAX && t && n && g && grad, "mean_square_bwd: bad arguments (<= %d tensors)", MS_MAX);
  for (int i = 0; i < k; ++i) VR_CHECK_ARG(t[i] && grad[i] && n[i] > 0, "mean_square_bwd: bad tensor %d", i);
  MsTable tb{};
  long nwg;
  ms_table(k, t, grad, n, &tb, &nwg);
  VR_CHECK_ARG(nwg < (1L << 31), "mean_square_bwd: too large");
  hipLaunchKernelGGL(mean_square_grad_kernel, dim3((unsigned)nwg), dim3(256), 0, vr_stream(stream), tb, g);
  VR_LAUNCH_CHECK("mean_square_bwd");
  return VR_OK;
}
