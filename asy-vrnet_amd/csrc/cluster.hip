// Context-Cluster core (the token mixer between fc1/fc_v and fc2), forward and backward.
// Reference: Cluster.forward backbone/fusion/vr_coc.py:158-190 (fold, 2x2 adaptive-avg-pool centre
// proposals, cosine similarity :114-125, sigmoid(beta + alpha*cos), hard argmax assignment, weighted
// aggregate, dispatch, unfold).  The reference materialises [b,4,N,D] broadcast products twice; here one
// workgroup owns one region-head (N points x D dims, M = 4 centres), keeps its feature / value points in
// registers (8 lanes per point, 4 dims per lane, float4 global accesses that cover whole 128-B lines at
// D = 32) and does every reduction with wave shuffles + one LDS hop.  HBM traffic is the algorithmic
// minimum: read f, v, write out (forward); read f, v, g, write df, dv (backward).
// All reductions have a fixed order: results are bitwise reproducible run to run.
#include "common.h"
#include <cstdlib>

namespace {

struct ClusterArgs {
  const float* f; const float* v; long ld;
  const float* alpha; const float* beta;
  const float* alpha2; const float* beta2;   // two-stream launch: samples >= B/2 use the second Cluster's similarity scale / shift
  float* out; long ldo;
  unsigned char* idx; float* wgt;
  // backward
  const float* g; long ldg;
  float* df; float* dv; long lddf;
  float* ab_partial;   // [blocks][2]
  int B, H, W, E, D, fold;
  int forced;          // forward: idx is an INPUT (teacher-forced assignment: parity tests), not computed here
  // optional bf16-plane copies (csrc/pgemm.hip): forward `out`; backward [df | dv] as ONE tensor of 2 E D columns (df first)
  vrnet_planes_out outp, dfvp;
  int in_bf16;         // f, v (and, backward, g) are bf16 tensors (row strides in elements): compute_dtype "bf16" with bf16 tensors
  // Streaming kernel (regions of more than 256 points), round 4: the forward may leave, per region-head, the centres, the
  // aggregated values and the assignment counts (CL_STATE floats); given them and the forward's similarity map, the backward
  // starts at its third pass: f is read twice instead of four times, v once instead of three times.
  float* state; const float* wgt_fwd;
};
constexpr int CL_STATE = 264;      // [0,128) centres of f | [128,256) a_m = (sum w v + vc_m) / (cnt_m + 1) | [256,260) cnt_m
__device__ __forceinline__ f32x4 cl_ld4(const float* base, long off, int bf16) {
  if (!bf16) return *reinterpret_cast<const f32x4*>(base + off);
  const vr_bf16x4 v = *reinterpret_cast<const vr_bf16x4*>(reinterpret_cast<const unsigned short*>(base) + off);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void cl_planes(const vrnet_planes_out& o, long row, int col, const f32x4 v) {
  if (o.p) vr_store_planes4(reinterpret_cast<unsigned short*>(o.p) + row * o.ld + col, o.plane, o.np, v);
}

constexpr int MAXW = 16;   // waves per workgroup

// Cross-lane moves use DPP modifiers (one VALU op, no LDS crossbar trip as ds_bpermute needs):
//   0xB1 quad_perm[1,0,3,2] (= xor 1), 0x4E quad_perm[2,3,0,1] (= xor 2), 0x141 row_half_mirror (lane i <-> 7-i of
//   each 8), 0x140 row_mirror (i <-> 15-i of each 16), 0x128 row_ror:8 (= xor 8 within a row of 16).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}

// Sum vals[m*4+q] (this lane's 4 dims of centre m) over every point of the workgroup.
// Result: dst[m*32 + d] * scale.  part: [NW*4][128] scratch (one partial per 16-lane row).
__device__ __forceinline__ void reduce_md(float (&vals)[16], float* part, float* dst, float scale, int tid, int T) {
  const int lane = tid & 63, NR = T >> 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) vals[i] += dpp_f<0x128>(vals[i]);
  if ((lane & 15) < 8) {
    float* dstp = part + (tid >> 4) * 128 + 4 * (lane & 7);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      f32x4 o = {vals[m * 4], vals[m * 4 + 1], vals[m * 4 + 2], vals[m * 4 + 3]};
      *reinterpret_cast<f32x4*>(dstp + m * 32) = o;
    }
  }
  __syncthreads();
  for (int i = tid; i < 128; i += T) {
    float s = 0.f;
    for (int r = 0; r < NR; ++r) s += part[r * 128 + i];
    dst[i] = s * scale;
  }
  __syncthreads();
}

// Sum 4 per-lane scalars over the workgroup -> dst[0..3].  part: [NW*4][4] scratch.
__device__ __forceinline__ void reduce4(float (&vals)[4], float* part, float* dst, int tid, int T) {
  const int NR = T >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float x = vals[i];
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x141>(x);
    x += dpp_f<0x140>(x);
    vals[i] = x;
  }
  if ((tid & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) part[(tid >> 4) * 4 + i] = vals[i];
  }
  __syncthreads();
  if (tid < 4) {
    float s = 0.f;
    for (int r = 0; r < NR; ++r) s += part[r * 4 + tid];
    dst[tid] = s;
  }
  __syncthreads();
}

// Hard assignment of one point (vr_coc.py:171-176).  All 8 lanes of the point's group hold the 4 dot products dt[]
// and the point norm; lane (sub & 3) evaluates centre m = sub & 3 only (one sigmoid per lane instead of four),
// then the 4 candidates are combined with two shuffles: larger similarity wins, ties go to the LOWER centre
// index (torch.max(dim) returns the first maximum).  Returns k; wgt = similarity, cosk = cosine of centre k.
__device__ __forceinline__ int assign4(const float (&dt)[4], float inv_nf, const float (&inv_cn)[4], float alpha,
                                       float beta, int sub, float& wgt, float& cosk) {
  const int m = sub & 3;
  const float d = m == 0 ? dt[0] : (m == 1 ? dt[1] : (m == 2 ? dt[2] : dt[3]));
  const float ic = m == 0 ? inv_cn[0] : (m == 1 ? inv_cn[1] : (m == 2 ? inv_cn[2] : inv_cn[3]));
  float c = d * inv_nf * ic;
  float sg = vr_sigmoid(beta + alpha * c);
  int k = m;
#pragma unroll
  for (int o = 1; o <= 2; o <<= 1) {
    const float s2 = o == 1 ? dpp_f<0xB1>(sg) : dpp_f<0x4E>(sg), c2 = o == 1 ? dpp_f<0xB1>(c) : dpp_f<0x4E>(c);
    const int k2 = o == 1 ? dpp_i<0xB1>(k) : dpp_i<0x4E>(k);
    const bool take = s2 > sg || (s2 == sg && k2 < k);
    sg = take ? s2 : sg;
    c = take ? c2 : c;
    k = take ? k2 : k;
  }
  wgt = sg;
  cosk = c;
  return k;
}
// similarity of a GIVEN centre k (backward replay of the saved assignment)
__device__ __forceinline__ void sim_of(const float (&dt)[4], int k, float inv_nf, const float (&inv_cn)[4], float alpha,
                                       float beta, float& wgt, float& cosk) {
  const float d = k == 0 ? dt[0] : (k == 1 ? dt[1] : (k == 2 ? dt[2] : dt[3]));
  const float ic = k == 0 ? inv_cn[0] : (k == 1 ? inv_cn[1] : (k == 2 ? inv_cn[2] : inv_cn[3]));
  cosk = d * inv_nf * ic;
  wgt = vr_sigmoid(beta + alpha * cosk);
}

__device__ __forceinline__ float group8_sum(float x) {
  x += dpp_f<0xB1>(x);
  x += dpp_f<0x4E>(x);
  x += dpp_f<0x141>(x);   // both quads already hold their own sum, so the mirror partner supplies the other quad's
  return x;
}

// LDS carve (floats): part[MAXW*4*128] | cen[128] | vcen[128] | agg[128] | afin[128] | t1[128] | t2[128] | misc[32]
constexpr int SM_PART = 0, SM_CEN = MAXW * 4 * 128, SM_VCEN = SM_CEN + 128, SM_AGG = SM_VCEN + 128,
              SM_AFIN = SM_AGG + 128, SM_T1 = SM_AFIN + 128, SM_T2 = SM_T1 + 128, SM_MISC = SM_T2 + 128,
              SM_TOTAL = SM_MISC + 32;

template <int NPT, bool BWD, int MAXT>
__global__ __launch_bounds__(MAXT) void cluster_kernel(const ClusterArgs p) {
  __shared__ float sm[SM_TOTAL];
  const int T = blockDim.x, tid = threadIdx.x, sub = tid & 7, pip = tid >> 3, PP = T >> 3;
  int rid = blockIdx.x;
  const int fold = p.fold;
  const int f2 = rid % fold; rid /= fold;
  const int f1 = rid % fold; rid /= fold;
  const int e = rid % p.E;
  const int b = rid / p.E;
  const int h = p.H / fold, w = p.W / fold, N = h * w;
  const int y0 = f1 * h, x0 = f2 * w;
  const int D = p.D;
  const bool dim_ok = 4 * sub < D;
  const int hh = (h + 1) / 2, hl = h / 2, wh = (w + 1) / 2, wl = w / 2;
  const float invq = 1.f / (float)(hh * wh);
  const bool second = p.alpha2 != nullptr && 2 * b >= p.B;
  const float alpha = second ? p.alpha2[0] : p.alpha[0], beta = second ? p.beta2[0] : p.beta[0];

  float f[NPT][4], v[NPT][4];
  long row[NPT];
  bool ok[NPT];
  unsigned inq[NPT];   // bit m set: point lies in pooling window m
#pragma unroll
  for (int s = 0; s < NPT; ++s) {
    const int n = s * PP + pip;
    ok[s] = n < N;
    const int i = ok[s] ? n / w : 0, j = ok[s] ? n - i * w : 0;
    row[s] = ((long)(b * p.H + y0 + i) * p.W + x0 + j);
    const unsigned r0 = i < hh, r1 = i >= hl, c0 = j < wh, c1 = j >= wl;
    inq[s] = ok[s] ? ((r0 & c0) | ((r0 & c1) << 1) | ((r1 & c0) << 2) | ((r1 & c1) << 3)) : 0u;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
    if (ok[s] && dim_ok) {
      a = cl_ld4(p.f, row[s] * p.ld + e * D + 4 * sub, p.in_bf16);
      c = cl_ld4(p.v, row[s] * p.ld + e * D + 4 * sub, p.in_bf16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f[s][q] = a[q];
      v[s][q] = c[q];
    }
  }

  // ---- centres of f and v (2x2 adaptive average pooling over the region)
  {
    float cs[16], vs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cs[i] = vs[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NPT; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float pw = (inq[s] >> m) & 1u ? 1.f : 0.f;   // 0/1 weight instead of a branch: straight-line FMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cs[m * 4 + q] += pw * f[s][q];
          vs[m * 4 + q] += pw * v[s][q];
        }
      }
    reduce_md(cs, sm + SM_PART, sm + SM_CEN, invq, tid, T);
    reduce_md(vs, sm + SM_PART, sm + SM_VCEN, invq, tid, T);
  }
  float cl[4][4], cnorm[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cl[m][q] = sm[SM_CEN + m * 32 + 4 * sub + q];
      s2 = __builtin_fmaf(cl[m][q], cl[m][q], s2);
    }
    cnorm[m] = fmaxf(sqrtf(group8_sum(s2)), 1e-12f);
  }
  float inv_cn[4], chat[4][4];   // 1/|c_m| and the unit centres c_m/|c_m|
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    inv_cn[m] = 1.f / cnorm[m];
#pragma unroll
    for (int q = 0; q < 4; ++q) chat[m][q] = cl[m][q] * inv_cn[m];
  }

  // ---- similarity, hard assignment
  float wg[NPT], fn[NPT], cosk[NPT];
  int kk[NPT];
#pragma unroll
  for (int s = 0; s < NPT; ++s) {
    float n2 = 0.f, dt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      n2 = __builtin_fmaf(f[s][q], f[s][q], n2);
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] = __builtin_fmaf(f[s][q], cl[m][q], dt[m]);   // explicit: identical centres must
                                                                                    // give bit-identical dots (ties)
    }
    n2 = group8_sum(n2);
#pragma unroll
    for (int m = 0; m < 4; ++m) dt[m] = group8_sum(dt[m]);
    const float nf = fmaxf(sqrtf(n2), 1e-12f);
    const float inv_nf = 1.f / nf;
    fn[s] = inv_nf;   // 1/max(|f_n|, eps); the clamp is active iff inv_nf >= 1e12
    float best, bc;
    int k;
    if (BWD || p.forced) {
      k = ok[s] ? (int)p.idx[row[s] * p.E + e] : 0;     // replay the forward's (or a given) assignment
      sim_of(dt, k, inv_nf, inv_cn, alpha, beta, best, bc);
    } else {
      k = assign4(dt, inv_nf, inv_cn, alpha, beta, sub, best, bc);
    }
    wg[s] = best;
    kk[s] = k;
    cosk[s] = bc;
  }

  // ---- aggregate: a_m = (sum_{k(n)=m} w_n v_n + vc_m) / (cnt_m + 1)
  {
    float ag[16], cnt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) ag[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NPT; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool mine = ok[s] && kk[s] == m;
        const float wk = mine ? wg[s] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) ag[m * 4 + q] += wk * v[s][q];
        cnt[m] += (mine && sub == 0) ? 1.f : 0.f;
      }
    reduce_md(ag, sm + SM_PART, sm + SM_AGG, 1.f, tid, T);
    reduce4(cnt, sm + SM_PART, sm + SM_MISC, tid, T);
    for (int i = tid; i < 128; i += T) sm[SM_AFIN + i] = (sm[SM_AGG + i] + sm[SM_VCEN + i]) / (sm[SM_MISC + (i >> 5)] + 1.f);
    __syncthreads();
  }

  if (!BWD) {
#pragma unroll
    for (int s = 0; s < NPT; ++s) {
      if (!ok[s]) continue;
      if (dim_ok) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = wg[s] * sm[SM_AFIN + kk[s] * 32 + 4 * sub + q];
        if (p.out) *reinterpret_cast<f32x4*>(p.out + row[s] * p.ldo + e * D + 4 * sub) = o;
        cl_planes(p.outp, row[s], e * D + 4 * sub, o);
      }
      if (sub == 0) {
        p.idx[row[s] * p.E + e] = (unsigned char)kk[s];
        if (p.wgt) p.wgt[row[s] * p.E + e] = wg[s];
      }
    }
    return;
  }

  // ================================ backward ================================
  float g[NPT][4];
#pragma unroll
  for (int s = 0; s < NPT; ++s) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok[s] && dim_ok) a = cl_ld4(p.g, row[s] * p.ldg + e * D + 4 * sub, p.in_bf16);
#pragma unroll
    for (int q = 0; q < 4; ++q) g[s][q] = a[q];
  }
  // da_m = sum_{k(n)=m} w_n g_n ;  at_m = da_m / (cnt_m + 1)   -> SM_T1
  {
    float da[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) da[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NPT; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float wk = (ok[s] && kk[s] == m) ? wg[s] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) da[m * 4 + q] += wk * g[s][q];
      }
    reduce_md(da, sm + SM_PART, sm + SM_T1, 1.f, tid, T);
    for (int i = tid; i < 128; i += T) sm[SM_T1 + i] = sm[SM_T1 + i] / (sm[SM_MISC + (i >> 5)] + 1.f);
    __syncthreads();
  }
  // per point: dw, dz, dcos; dv; accumulate d c_hat
  float dcs[NPT];
  float dal = 0.f, dbe = 0.f;
  {
    float dch[16], atq[4][4];   // atq: this lane's 4 dims of at_m, all 4 centres
#pragma unroll
    for (int i = 0; i < 16; ++i) dch[i] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) atq[m][q] = sm[SM_T1 + m * 32 + 4 * sub + q];
#pragma unroll
    for (int s = 0; s < NPT; ++s) {
      const int k = kk[s];     // padded points (!ok) run the same straight-line code with zero weights
      float at[4], ak[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        at[q] = sm[SM_T1 + k * 32 + 4 * sub + q];
        ak[q] = sm[SM_AFIN + k * 32 + 4 * sub + q];
      }
      float dwp = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) dwp += g[s][q] * ak[q] + at[q] * v[s][q];
      const float dw = group8_sum(dwp);
      const float dz = dw * wg[s] * (1.f - wg[s]);
      const float live = (ok[s] && sub == 0) ? 1.f : 0.f;
      dbe += live * dz;
      dal += live * dz * cosk[s];
      const float dc = ok[s] ? alpha * dz : 0.f;
      dcs[s] = dc;
      // dv_n = w_n * at_k + sum_{m: n in Q_m} at_m / |Q_m|
      {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = wg[s] * at[q];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float pw = (inq[s] >> m) & 1u ? invq : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] += pw * atq[m][q];
        }
        if (ok[s] && dim_ok) {
          if (p.dv) *reinterpret_cast<f32x4*>(p.dv + row[s] * p.lddf + e * D + 4 * sub) = o;
          cl_planes(p.dfvp, row[s], p.E * D + e * D + 4 * sub, o);
        }
      }
      // d c_hat_k += dcos * f_hat_n
      const float dcn = dc * fn[s];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float wk = k == m ? dcn : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) dch[m * 4 + q] += wk * f[s][q];
      }
    }
    reduce_md(dch, sm + SM_PART, sm + SM_T2, 1.f, tid, T);
  }
  // dc_m = (dch_m - chat_m (chat_m . dch_m)) / max(|c_m|, eps)   [no projection when |c_m| < eps]
  float dcen[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float dh[4], dot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dh[q] = sm[SM_T2 + m * 32 + 4 * sub + q];
      dot += chat[m][q] * dh[q];
    }
    dot = group8_sum(dot);
    const bool clamped = cnorm[m] <= 1e-12f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dcen[m][q] = clamped ? dh[q] * inv_cn[m] : (dh[q] - chat[m][q] * dot) * inv_cn[m];
  }
#pragma unroll
  for (int s = 0; s < NPT; ++s) {
    const int k = kk[s];
    const float inv_nf = fn[s];
    const bool clamped = inv_nf >= 1e12f;
    const float dn = dcs[s] * inv_nf, proj = clamped ? 0.f : inv_nf * cosk[s];
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ck = k == 0 ? chat[0][q] : (k == 1 ? chat[1][q] : (k == 2 ? chat[2][q] : chat[3][q]));
      o[q] = dn * (ck - f[s][q] * proj);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float pw = (inq[s] >> m) & 1u ? invq : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] += pw * dcen[m][q];
    }
    if (ok[s] && dim_ok) {
      if (p.df) *reinterpret_cast<f32x4*>(p.df + row[s] * p.lddf + e * D + 4 * sub) = o;
      cl_planes(p.dfvp, row[s], e * D + 4 * sub, o);
    }
  }
  // d alpha, d beta partials of this workgroup
  {
    float ab[4] = {dal, dbe, 0.f, 0.f};
    reduce4(ab, sm + SM_PART, sm + SM_MISC + 8, tid, T);
    if (tid == 0) {
      p.ab_partial[2 * (long)blockIdx.x] = sm[SM_MISC + 8];
      p.ab_partial[2 * (long)blockIdx.x + 1] = sm[SM_MISC + 9];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming variant for regions that do not fit the register-resident kernel (N > 256 points: neck p3 at
// 512 px, every backbone stage at 1024 px, neck p3 at 1024 px with N = 4096).  Same arithmetic, same
// reduction order per chunk; the points are re-read from L2/HBM in chunks of 64 (512 threads, 8 lanes per
// point) instead of being held in registers: forward reads f, v twice; backward reads f 4x, v 3x, g 2x and
// keeps one float per point (d cos) in a scratch map.  p.wgt: forward = similarity map (required),
// backward = the per-point scratch map.
template <bool BWD>
__global__ __launch_bounds__(512) void cluster_stream_kernel(const ClusterArgs p) {
  __shared__ float sm[SM_TOTAL];
  const int T = 512, tid = threadIdx.x, sub = tid & 7, pip = tid >> 3, PP = T >> 3;
  int rid = blockIdx.x;
  const int fold = p.fold;
  const int f2 = rid % fold; rid /= fold;
  const int f1 = rid % fold; rid /= fold;
  const int e = rid % p.E;
  const int b = rid / p.E;
  const int h = p.H / fold, w = p.W / fold, N = h * w;
  const int y0 = f1 * h, x0 = f2 * w;
  const int D = p.D;
  const bool dim_ok = 4 * sub < D;
  const int hh = (h + 1) / 2, hl = h / 2, wh = (w + 1) / 2, wl = w / 2;
  const float invq = 1.f / (float)(hh * wh);
  const bool second = p.alpha2 != nullptr && 2 * b >= p.B;
  const float alpha = second ? p.alpha2[0] : p.alpha[0], beta = second ? p.beta2[0] : p.beta[0];
  const int coff = e * D + 4 * sub;

  auto locate = [&](int n, long& row, unsigned& inq) -> bool {
    const bool ok = n < N;
    const int i = ok ? n / w : 0, j = ok ? n - i * w : 0;
    row = ((long)(b * p.H + y0 + i) * p.W + x0 + j);
    const unsigned r0 = i < hh, r1 = i >= hl, c0 = j < wh, c1 = j >= wl;
    inq = ok ? ((r0 & c0) | ((r0 & c1) << 1) | ((r1 & c0) << 2) | ((r1 & c1) << 3)) : 0u;
    return ok;
  };
  auto load4 = [&](const float* base, long ld, long row, bool ok, float (&o)[4]) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (ok && dim_ok) a = cl_ld4(base, row * ld + coff, p.in_bf16);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = a[q];
  };

  // backward with the forward's saved state: centres, aggregates and counts come from `state`, the weights from the forward's
  // similarity map -- passes 1 and 2 shrink to one pass over g (da = sum w g); the values are the ones those passes would
  // recompute, bit for bit (tests/test_hip_ops.py::test_cluster_core)
  const bool saved = BWD && p.state != nullptr && p.wgt_fwd != nullptr;
  if (saved) {
    const float* stt = p.state + (long)blockIdx.x * CL_STATE;
    for (int i = tid; i < 128; i += T) {
      sm[SM_CEN + i] = stt[i];
      sm[SM_AFIN + i] = stt[128 + i];
      if (i < 4) sm[SM_MISC + i] = stt[256 + i];
    }
    __syncthreads();
  }
  // ---- pass 1: centres of f and v
  if (!saved) {
    float cs[16], vs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cs[i] = vs[i] = 0.f;
#pragma unroll(BWD ? 1 : 4)      // forward: four points' loads in flight per lane group (1.50 -> 1.41 ms per step at 1 024 px); the backward spills with it
    for (int n0 = 0; n0 < N; n0 += PP) {
      long row; unsigned inq;
      const bool ok = locate(n0 + pip, row, inq);
      float f[4], v[4];
      load4(p.f, p.ld, row, ok, f);
      load4(p.v, p.ld, row, ok, v);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float pw = (inq >> m) & 1u ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { cs[m * 4 + q] += pw * f[q]; vs[m * 4 + q] += pw * v[q]; }
      }
    }
    reduce_md(cs, sm + SM_PART, sm + SM_CEN, invq, tid, T);
    reduce_md(vs, sm + SM_PART, sm + SM_VCEN, invq, tid, T);
  }
  float cl[4][4], cnorm[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { cl[m][q] = sm[SM_CEN + m * 32 + 4 * sub + q]; s2 = __builtin_fmaf(cl[m][q], cl[m][q], s2); }
    cnorm[m] = fmaxf(sqrtf(group8_sum(s2)), 1e-12f);
  }
  float inv_cn[4], chat[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    inv_cn[m] = 1.f / cnorm[m];
#pragma unroll
    for (int q = 0; q < 4; ++q) chat[m][q] = cl[m][q] * inv_cn[m];
  }
  // similarity of one point (all 8 lanes of its group get the same values)
  auto assign = [&](const float (&f)[4], bool ok, long row, float& nf, float& wgt, float& cosk) -> int {
    float n2 = 0.f, dt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      n2 = __builtin_fmaf(f[q], f[q], n2);
#pragma unroll
      for (int m = 0; m < 4; ++m) dt[m] = __builtin_fmaf(f[q], cl[m][q], dt[m]);
    }
    n2 = group8_sum(n2);
#pragma unroll
    for (int m = 0; m < 4; ++m) dt[m] = group8_sum(dt[m]);
    const float inv_nf = 1.f / fmaxf(sqrtf(n2), 1e-12f);
    nf = inv_nf;   // callers get 1/max(|f_n|, eps)
    float best, bc;
    int k;
    if (BWD || p.forced) {
      k = ok ? (int)p.idx[row * p.E + e] : 0;
      sim_of(dt, k, inv_nf, inv_cn, alpha, beta, best, bc);
    } else {
      k = assign4(dt, inv_nf, inv_cn, alpha, beta, sub, best, bc);
    }
    wgt = best; cosk = bc;
    return k;
  };

  if (saved) {
    float da[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) da[i] = 0.f;
#pragma unroll 4
    for (int n0 = 0; n0 < N; n0 += PP) {
      long row; unsigned inq;
      const bool ok = locate(n0 + pip, row, inq);
      float g[4];
      load4(p.g, p.ldg, row, ok, g);
      const int k = ok ? (int)p.idx[row * p.E + e] : 0;
      const float wg = ok ? p.wgt_fwd[row * p.E + e] : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float wk = (ok && k == m) ? wg : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) da[m * 4 + q] += wk * g[q];
      }
    }
    reduce_md(da, sm + SM_PART, sm + SM_T1, 1.f, tid, T);
    for (int i = tid; i < 128; i += T) sm[SM_T1 + i] = sm[SM_T1 + i] / (sm[SM_MISC + (i >> 5)] + 1.f);
    __syncthreads();
  }
  // ---- pass 2: assignment, aggregate (and, backward, da = sum w g)
  if (!saved) {
  {
    float ag[16], da[16], cnt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) ag[i] = da[i] = 0.f;
#pragma unroll(BWD ? 1 : 4)      // forward: four points' loads in flight per lane group (1.50 -> 1.41 ms per step at 1 024 px); the backward spills with it
    for (int n0 = 0; n0 < N; n0 += PP) {
      long row; unsigned inq;
      const bool ok = locate(n0 + pip, row, inq);
      float f[4], v[4], g[4], nf, wg, ck;
      load4(p.f, p.ld, row, ok, f);
      load4(p.v, p.ld, row, ok, v);
      if (BWD) load4(p.g, p.ldg, row, ok, g);
      const int k = assign(f, ok, row, nf, wg, ck);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool mine = ok && k == m;
        const float wk = mine ? wg : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ag[m * 4 + q] += wk * v[q]; if (BWD) da[m * 4 + q] += wk * g[q]; }
        cnt[m] += (mine && sub == 0) ? 1.f : 0.f;
      }
      if (!BWD && ok && sub == 0) {
        p.idx[row * p.E + e] = (unsigned char)k;
        p.wgt[row * p.E + e] = wg;
      }
    }
    reduce_md(ag, sm + SM_PART, sm + SM_AGG, 1.f, tid, T);
    reduce4(cnt, sm + SM_PART, sm + SM_MISC, tid, T);
    if (BWD) reduce_md(da, sm + SM_PART, sm + SM_T1, 1.f, tid, T);
    for (int i = tid; i < 128; i += T) {
      const float den = sm[SM_MISC + (i >> 5)] + 1.f;
      sm[SM_AFIN + i] = (sm[SM_AGG + i] + sm[SM_VCEN + i]) / den;
      if (BWD) sm[SM_T1 + i] = sm[SM_T1 + i] / den;
      if (!BWD && p.state) {
        float* stt = p.state + (long)blockIdx.x * CL_STATE;
        stt[i] = sm[SM_CEN + i];
        stt[128 + i] = sm[SM_AFIN + i];
        if (i < 4) stt[256 + i] = sm[SM_MISC + i];
      }
    }
    __syncthreads();
  }
  }

  if (!BWD) {   // ---- pass 3: dispatch
#pragma unroll(BWD ? 1 : 4)      // forward: four points' loads in flight per lane group (1.50 -> 1.41 ms per step at 1 024 px); the backward spills with it
    for (int n0 = 0; n0 < N; n0 += PP) {
      long row; unsigned inq;
      const bool ok = locate(n0 + pip, row, inq);
      if (ok && dim_ok) {
        const int k = p.idx[row * p.E + e];
        const float wg = p.wgt[row * p.E + e];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = wg * sm[SM_AFIN + k * 32 + 4 * sub + q];
        if (p.out) *reinterpret_cast<f32x4*>(p.out + row * p.ldo + coff) = o;
        cl_planes(p.outp, row, coff, o);
      }
    }
    return;
  }

  // ---- backward pass 3: dw, dz, d cos (to scratch), dv, d c_hat
  float dal = 0.f, dbe = 0.f;
  {
    float dch[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dch[i] = 0.f;
#pragma unroll(BWD ? 1 : 4)      // forward: four points' loads in flight per lane group (1.50 -> 1.41 ms per step at 1 024 px); the backward spills with it
    for (int n0 = 0; n0 < N; n0 += PP) {
      long row; unsigned inq;
      const bool ok = locate(n0 + pip, row, inq);
      float f[4], v[4], g[4], nf, wg, ck;
      load4(p.f, p.ld, row, ok, f);
      load4(p.v, p.ld, row, ok, v);
      load4(p.g, p.ldg, row, ok, g);
      const int k = assign(f, ok, row, nf, wg, ck);
      float at[4], ak[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        at[q] = sm[SM_T1 + k * 32 + 4 * sub + q];
        ak[q] = sm[SM_AFIN + k * 32 + 4 * sub + q];
      }
      float dwp = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) dwp += g[q] * ak[q] + at[q] * v[q];
      const float dw = group8_sum(dwp);
      const float dz = dw * wg * (1.f - wg);
      const float dc = alpha * dz;
      if (ok) {
        if (sub == 0) {
          dbe += dz;
          dal += dz * ck;
          p.wgt[row * p.E + e] = dc;
        }
        if (dim_ok) {
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = wg * at[q];
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (inq & (1u << m)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] += invq * sm[SM_T1 + m * 32 + 4 * sub + q];
            }
          if (p.dv) *reinterpret_cast<f32x4*>(p.dv + row * p.lddf + coff) = o;
          cl_planes(p.dfvp, row, p.E * p.D + coff, o);
        }
        const float dcn = dc * nf;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float wk = k == m ? dcn : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) dch[m * 4 + q] += wk * f[q];
        }
      }
    }
    reduce_md(dch, sm + SM_PART, sm + SM_T2, 1.f, tid, T);
  }
  float dcen[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float dh[4], dot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { dh[q] = sm[SM_T2 + m * 32 + 4 * sub + q]; dot += chat[m][q] * dh[q]; }
    dot = group8_sum(dot);
    const bool clamped = cnorm[m] <= 1e-12f;
#pragma unroll
    for (int q = 0; q < 4; ++q) dcen[m][q] = clamped ? dh[q] * inv_cn[m] : (dh[q] - chat[m][q] * dot) * inv_cn[m];
  }
  // ---- backward pass 4: df
#pragma unroll(BWD ? 1 : 4)
  for (int n0 = 0; n0 < N; n0 += PP) {
    long row; unsigned inq;
    const bool ok = locate(n0 + pip, row, inq);
    float f[4], nf, wg, ck;
    load4(p.f, p.ld, row, ok, f);
    const int k = assign(f, ok, row, nf, wg, ck);
    if (!ok || !dim_ok) continue;
    const float dc = p.wgt[row * p.E + e];
    const float inv_nf = nf;
    const bool clamped = inv_nf >= 1e12f;
    const float dn = dc * inv_nf, proj = clamped ? 0.f : inv_nf * ck;
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ckq = k == 0 ? chat[0][q] : (k == 1 ? chat[1][q] : (k == 2 ? chat[2][q] : chat[3][q]));
      o[q] = dn * (ckq - f[q] * proj);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (inq & (1u << m)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] += invq * dcen[m][q];
      }
    if (p.df) *reinterpret_cast<f32x4*>(p.df + row * p.lddf + coff) = o;
    cl_planes(p.dfvp, row, coff, o);
  }
  {
    float ab[4] = {dal, dbe, 0.f, 0.f};
    reduce4(ab, sm + SM_PART, sm + SM_MISC + 8, tid, T);
    if (tid == 0) {
      p.ab_partial[2 * (long)blockIdx.x] = sm[SM_MISC + 8];
      p.ab_partial[2 * (long)blockIdx.x + 1] = sm[SM_MISC + 9];
    }
  }
}

__global__ __launch_bounds__(256) void cluster_ab_reduce_kernel(const float* partial, long blocks, float* dalpha,
                                                                float* dbeta, int accumulate, float* dalpha2, float* dbeta2) {
  __shared__ double red[8];
  if (blockIdx.x) {        // second stream: the second half of the workgroups (region ids are sample-major)
    partial += 2 * blocks;
    dalpha = dalpha2; dbeta = dbeta2;
  }
  double a = 0, b = 0;
  for (long i = threadIdx.x; i < blocks; i += 256) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = a;
    red[4 + (threadIdx.x >> 6)] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sa = red[0] + red[1] + red[2] + red[3], sb = red[4] + red[5] + red[6] + red[7];
    dalpha[0] = (accumulate ? dalpha[0] : 0.f) + (float)sa;
    dbeta[0] = (accumulate ? dbeta[0] : 0.f) + (float)sb;
  }
}

// The same reduction for up to AB_MAX Cluster modules in ONE launch (round 5): the backward kernels of a section leave their
// per-workgroup (d alpha, d beta) partials in buffers the caller owns, and one workgroup per module adds them up once the
// section's chains have joined -- 27 launches per step off the backward chains.  The table travels by value (kernel
// arguments), so a captured graph needs no host memory.
constexpr int AB_MAX = 32;
struct AbTable {
  const float* partial[AB_MAX];
  float* dalpha[AB_MAX];
  float* dbeta[AB_MAX];
  int blocks[AB_MAX];
  int accumulate[AB_MAX];
};
__global__ __launch_bounds__(256) void cluster_ab_reduce_multi_kernel(const AbTable t) {
  __shared__ double red[8];
  const int e = blockIdx.x;
  const float* partial = t.partial[e];
  const long blocks = t.blocks[e];
  double a = 0, b = 0;
  for (long i = threadIdx.x; i < blocks; i += 256) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = a;
    red[4 + (threadIdx.x >> 6)] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sa = red[0] + red[1] + red[2] + red[3], sb = red[4] + red[5] + red[6] + red[7];
    t.dalpha[e][0] = (t.accumulate[e] ? t.dalpha[e][0] : 0.f) + (float)sa;
    t.dbeta[e][0] = (t.accumulate[e] ? t.dbeta[e][0] : 0.f) + (float)sb;
  }
}

int cluster_plan(int N, long blocks, int* T, int* npt, int bwd) {
  if (N > 256) {      // streaming kernel (points re-read in chunks): any region size
    *T = 0;
    *npt = 0;
    return 0;
  }
  int t = ((N + 63) / 64) * 64;
  if (t < 64) t = 64;
  // few region-heads (stages 2-3: B*E*fold^2 = 256 / 64 workgroups): spread each region over 4x the threads
  // (2 points per thread instead of 8) so that the chip holds 4x the waves and loads in flight
  if (blocks <= 256 && N > 4 * (t / 8)) t *= 4;     // measured: 512 workgroups of 256 threads already do better as they are
  // forward, many region-heads of 256 points: 512 threads x 4 points (110 VGPRs, 16 waves per CU) instead of 256 x 8
  // (183 VGPRs, 8 waves per CU): +8 % (stage 0: 54.2 -> 49.9 us).  The backward kernel would spill at 128 VGPRs: it stays.
  if (!bwd && blocks > 256 && t == 256 && N > 128) t = 512;
  static const int force_t = vr_tune("VRNET_CLUSTER_T", 0);      // tuning aid
  if (force_t >= 64 && force_t % 64 == 0 && blocks > 256) t = force_t;
  if (t > 1024) t = 1024;
  const int pp = t / 8;
  const int np = (N + pp - 1) / pp;
  if (np > 8) return -1;
  *T = t;
  *npt = np <= 1 ? 1 : (np <= 2 ? 2 : (np <= 4 ? 4 : 8));
  return 0;
}

template <bool BWD>
int cluster_launch(const ClusterArgs& p, int T, int npt, long blocks, hipStream_t st) {
  if (T == 0) {
    hipLaunchKernelGGL((cluster_stream_kernel<BWD>), dim3(blocks), dim3(512), 0, st, p);
    return 0;
  }
  dim3 grid(blocks), block(T);
  if (T <= 256) {   // registers: up to 512 / (T/256) per lane; the 1024-thread variant is capped at 128
    switch (npt) {
      case 1: hipLaunchKernelGGL((cluster_kernel<1, BWD, 256>), grid, block, 0, st, p); break;
      case 2: hipLaunchKernelGGL((cluster_kernel<2, BWD, 256>), grid, block, 0, st, p); break;
      case 4: hipLaunchKernelGGL((cluster_kernel<4, BWD, 256>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((cluster_kernel<8, BWD, 256>), grid, block, 0, st, p); break;
    }
  } else {
    switch (npt) {
      case 1: hipLaunchKernelGGL((cluster_kernel<1, BWD, 1024>), grid, block, 0, st, p); break;
      case 2: hipLaunchKernelGGL((cluster_kernel<2, BWD, 1024>), grid, block, 0, st, p); break;
      case 4: hipLaunchKernelGGL((cluster_kernel<4, BWD, 1024>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((cluster_kernel<8, BWD, 1024>), grid, block, 0, st, p); break;
    }
  }
  return 0;
}

int cluster_check(const char* name, const void* f, const void* v, long ld, int B, int H, int W, int E, int D, int fold,
                  int* T, int* npt, int bwd) {
  VR_CHECK_ARG(f && v, "%s: null tensor", name);
  VR_CHECK_ARG(B > 0 && H > 0 && W > 0 && E > 0 && fold >= 1, "%s: bad shape", name);
  VR_CHECK_ARG(D > 0 && D <= 32 && D % 4 == 0, "%s: head_dim %d unsupported (multiple of 4, <= 32)", name, D);
  VR_CHECK_ARG(H % fold == 0 && W % fold == 0,
               "Ensure the feature map size (%d*%d) can be divided by fold %d*%d", H, W, fold, fold);
  VR_CHECK_ARG(ld % 4 == 0 && vr_aligned16(f) && vr_aligned16(v), "%s: rows must be 16-byte aligned", name);
  const int N = (H / fold) * (W / fold);
  VR_CHECK_ARG(cluster_plan(N, (long)B * E * fold * fold, T, npt, bwd) == 0, "%s: unsupported region of %d points", name, N);
  return VR_OK;
}

}  // namespace

static int cluster_fwd_impl(const float* f, const float* v, long ld, const float* alpha, const float* beta, float* out, long ldo,
                            unsigned char* idx, float* wgt, int B, int H, int W, int E, int D, int fold, const float* alpha2,
                            const float* beta2, int forced, const vrnet_planes_out* outp, void* stream, float* state = nullptr);

extern "C" int vrnet_cluster_fwd_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                                     float* out, long ldo, unsigned char* idx, float* wgt, int B, int H, int W, int E,
                                     int D, int fold, const float* alpha2, const float* beta2, void* stream) {
  return cluster_fwd_impl(f, v, ld, alpha, beta, out, ldo, idx, wgt, B, H, W, E, D, fold, alpha2, beta2, 0, nullptr, stream);
}
/* The same with a second copy of `out` as bf16 planes (the A operand of the proj conv's plane GEMM and the x operand of its
 * weight gradient); forced != 0: the teacher-forced form. */
extern "C" int vrnet_cluster_fwd_planes_f32(const void* f, const void* v, long ld, int in_bf16, const float* alpha,
                                            const float* beta, float* out, long ldo, unsigned char* idx, float* wgt, int B,
                                            int H, int W, int E, int D, int fold, int forced, const vrnet_planes_out* outp,
                                            float* state, void* stream) {
  return cluster_fwd_impl(reinterpret_cast<const float*>(f), reinterpret_cast<const float*>(v), ld, alpha, beta, out, ldo, idx, wgt,
                          B, H, W, E, D, fold, nullptr, nullptr, (forced ? 1 : 0) | (in_bf16 ? 2 : 0), outp, stream, state);
}

/* Floats of per-region state the forward of vrnet_cluster_fwd_planes_f32 leaves in `state` for regions of more than 256
 * points (0: the region kernel keeps its points in registers and has no use for it). */
extern "C" long vrnet_cluster_state_floats(int B, int H, int W, int E, int fold) {
  if (B <= 0 || E <= 0 || fold <= 0 || H % fold || W % fold) return 0;
  int T, npt;
  if (cluster_plan((H / fold) * (W / fold), (long)B * E * fold * fold, &T, &npt, 1) != 0 || T != 0) return 0;
  return (long)B * E * fold * fold * CL_STATE;
}

/* The same forward with the hard assignment GIVEN (idx is read, not written): every point goes to the centre idx names,
 * with that centre's similarity.  For parity work: two arithmetic paths can only be compared to rounding level when the
 * numerically tied points of the arg-max (vr_coc.py:173-176) are decided the same way in both. */
extern "C" int vrnet_cluster_fwd_forced_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                                            float* out, long ldo, const unsigned char* idx, float* wgt, int B, int H, int W,
                                            int E, int D, int fold, const float* alpha2, const float* beta2, void* stream) {
  return cluster_fwd_impl(f, v, ld, alpha, beta, out, ldo, const_cast<unsigned char*>(idx), wgt, B, H, W, E, D, fold, alpha2,
                          beta2, 1, nullptr, stream);
}

static int cluster_fwd_impl(const float* f, const float* v, long ld, const float* alpha, const float* beta, float* out, long ldo,
                            unsigned char* idx, float* wgt, int B, int H, int W, int E, int D, int fold, const float* alpha2,
                            const float* beta2, int forced, const vrnet_planes_out* outp, void* stream, float* state) {
  if (vr_ablated("cluster")) return VR_OK;
  int T, npt;
  int rc = cluster_check("cluster_fwd", f, v, ld, B, H, W, E, D, fold, &T, &npt, 0);
  if (rc) return rc;
  VR_CHECK_ARG((out || outp) && idx && alpha && beta && (!out || (ldo % 4 == 0 && vr_aligned16(out))) && vr_planes_out_ok(outp, E * D),
               "cluster_fwd: bad output (the fp32 output may be NULL only with a plane output)");
  VR_CHECK_ARG(T != 0 || wgt, "cluster_fwd: regions of more than 256 points need the similarity map `wgt` (B,H,W,E)");
  VR_CHECK_ARG((!alpha2 == !beta2) && (!alpha2 || B % 2 == 0), "cluster_fwd: a two-stream launch needs alpha2, beta2 and an even batch");
  ClusterArgs p{};
  p.alpha2 = alpha2; p.beta2 = beta2;
  p.f = f; p.v = v; p.ld = ld; p.alpha = alpha; p.beta = beta; p.out = out; p.ldo = ldo; p.idx = idx; p.wgt = wgt;
  p.B = B; p.H = H; p.W = W; p.E = E; p.D = D; p.fold = fold;
  p.forced = forced & 1;
  p.in_bf16 = (forced >> 1) & 1;
  if (outp) p.outp = *outp;
  p.state = T == 0 ? state : nullptr;
  cluster_launch<false>(p, T, npt, (long)B * E * fold * fold, vr_stream(stream));
  VR_LAUNCH_CHECK("cluster_fwd");
  return VR_OK;
}

// partial (alpha, beta) sums per workgroup + (streaming kernel) one float per point and head
extern "C" long vrnet_cluster_bwd_workspace2(int B, int H, int W, int E, int fold) {
  return ((long)B * E * fold * fold * 2 + (long)B * H * W * E) * 4 + 512;
}
extern "C" long vrnet_cluster_bwd_workspace(int B, int E, int fold) { return (long)B * E * fold * fold * 2 * 4 + 256; }

static int cluster_bwd_impl(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                            const unsigned char* idx, const float* dout, long lddo, float* df, float* dv,
                            long lddf, float* dalpha, float* dbeta, int accumulate_ab, int B, int H, int W,
                            int E, int D, int fold, const float* alpha2, const float* beta2, float* dalpha2,
                            float* dbeta2, const vrnet_planes_out* dfvp, void* workspace, long workspace_bytes, void* stream,
                            const float* wgt_fwd = nullptr, const float* state = nullptr);
extern "C" int vrnet_cluster_bwd_f32(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                                     const unsigned char* idx, const float* dout, long lddo, float* df, float* dv,
                                     long lddf, float* dalpha, float* dbeta, int accumulate_ab, int B, int H, int W,
                                     int E, int D, int fold, const float* alpha2, const float* beta2, float* dalpha2,
                                     float* dbeta2, void* workspace, long workspace_bytes, void* stream) {
  // (any non-zero accumulate_ab means "accumulate", as in ABI 5: bit 1 of the internal flag word is the bf16-input switch)
  return cluster_bwd_impl(f, v, ld, alpha, beta, idx, dout, lddo, df, dv, lddf, dalpha, dbeta, accumulate_ab ? 1 : 0, B, H, W, E, D, fold,
                          alpha2, beta2, dalpha2, dbeta2, nullptr, workspace, workspace_bytes, stream);
}
/* The same with a second copy of [df | dv] (one tensor of 2 E D columns, df first) as bf16 planes: the dy operand of the
 * fc1 | fc_v data- and weight-gradient plane GEMMs. */
extern "C" int vrnet_cluster_bwd_planes_f32(const void* f, const void* v, long ld, int in_bf16, const float* alpha,
                                            const float* beta, const unsigned char* idx, const void* dout, long lddo, float* df,
                                            float* dv, long lddf, float* dalpha, float* dbeta, int accumulate_ab, int B, int H,
                                            int W, int E, int D, int fold, const vrnet_planes_out* dfvp, const float* wgt_fwd,
                                            const float* state, void* workspace, long workspace_bytes, void* stream) {
  VR_CHECK_ARG(!wgt_fwd == !state, "cluster_bwd: the forward's similarity map and region state come together");
  return cluster_bwd_impl(reinterpret_cast<const float*>(f), reinterpret_cast<const float*>(v), ld, alpha, beta, idx,
                          reinterpret_cast<const float*>(dout), lddo, df, dv, lddf, dalpha, dbeta, (accumulate_ab ? 1 : 0) | (in_bf16 ? 2 : 0),
                          B, H, W, E, D, fold, nullptr, nullptr, nullptr, nullptr, dfvp, workspace, workspace_bytes, stream, wgt_fwd, state);
}
static int cluster_bwd_impl(const float* f, const float* v, long ld, const float* alpha, const float* beta,
                            const unsigned char* idx, const float* dout, long lddo, float* df, float* dv,
                            long lddf, float* dalpha, float* dbeta, int accumulate_ab, int B, int H, int W,
                            int E, int D, int fold, const float* alpha2, const float* beta2, float* dalpha2,
                            float* dbeta2, const vrnet_planes_out* dfvp, void* workspace, long workspace_bytes, void* stream,
                            const float* wgt_fwd, const float* state) {
  if (vr_ablated("cluster")) return VR_OK;
  const int in_bf16 = (accumulate_ab >> 1) & 1;      // (bit 1: set by vrnet_cluster_bwd_planes_f32 only)
  accumulate_ab &= 1;
  int T, npt;
  int rc = cluster_check("cluster_bwd", f, v, ld, B, H, W, E, D, fold, &T, &npt, 1);
  if (rc) return rc;
  VR_CHECK_ARG(idx && dout && ((df && dv) || (dfvp && !df && !dv)) && (!dalpha == !dbeta) && workspace,
               "cluster_bwd: null tensor (df / dv may both be NULL only with a plane output; dalpha and dbeta come together)");
  VR_CHECK_ARG(dalpha || !alpha2, "cluster_bwd: the deferred (d alpha, d beta) form is single-stream");
  VR_CHECK_ARG(lddo % 4 == 0 && vr_aligned16(dout) && (!df || (lddf % 4 == 0 && vr_aligned16(df) && vr_aligned16(dv))) &&
                   vr_planes_out_ok(dfvp, 2 * E * D),
               "cluster_bwd: rows must be 16-byte aligned");
  const long blocks = (long)B * E * fold * fold;
  const long need = T == 0 ? vrnet_cluster_bwd_workspace2(B, H, W, E, fold) : vrnet_cluster_bwd_workspace(B, E, fold);
  if (workspace_bytes < need) {
    vr_set_error("cluster_bwd: workspace %ld < %ld bytes", workspace_bytes, need);
    return VR_ERR_WORKSPACE;
  }
  VR_CHECK_ARG((!alpha2 == !beta2) && (!alpha2 == !dalpha2) && (!alpha2 == !dbeta2) && (!alpha2 || B % 2 == 0),
               "cluster_bwd: a two-stream launch needs alpha2, beta2, dalpha2, dbeta2 and an even batch");
  ClusterArgs p{};
  p.alpha2 = alpha2; p.beta2 = beta2;
  p.f = f; p.v = v; p.ld = ld; p.alpha = alpha; p.beta = beta; p.idx = const_cast<unsigned char*>(idx);
  p.g = dout; p.ldg = lddo; p.df = df; p.dv = dv; p.lddf = lddf;
  p.in_bf16 = in_bf16;
  if (dfvp) p.dfvp = *dfvp;
  p.ab_partial = reinterpret_cast<float*>(workspace);
  p.wgt = p.ab_partial + ((blocks * 2 + 63) / 64) * 64;       // streaming kernel: per-point d cos scratch
  if (T == 0 && wgt_fwd && state) { p.wgt_fwd = wgt_fwd; p.state = const_cast<float*>(state); }
  p.B = B; p.H = H; p.W = W; p.E = E; p.D = D; p.fold = fold;
  hipStream_t st = vr_stream(stream);
  cluster_launch<true>(p, T, npt, blocks, st);
  VR_LAUNCH_CHECK("cluster_bwd");
  // dalpha == dbeta == NULL: the per-workgroup partials stay in the first 2 * B * E * fold^2 floats of `workspace` (which the
  // caller then owns until it has run vrnet_cluster_ab_reduce_multi over it)
  if (!dalpha) return VR_OK;
  hipLaunchKernelGGL(cluster_ab_reduce_kernel, dim3(alpha2 ? 2 : 1), dim3(256), 0, st, p.ab_partial,
                     alpha2 ? blocks / 2 : blocks, dalpha, dbeta, accumulate_ab, dalpha2, dbeta2);
  VR_LAUNCH_CHECK("cluster_ab_reduce");
  return VR_OK;
}

/* (d alpha, d beta) of n Cluster modules from the per-workgroup partials their backward launches left (vrnet_cluster_bwd_*
 * with dalpha = dbeta = NULL): partial[i] = that launch's workspace, blocks[i] = its B * E * fold^2, accumulate[i] != 0 adds to
 * the gradients.  Host arrays; one launch per 32 modules.  Replaces the per-module finishing launch behind every
 * Cluster backward (backbone/fusion/vr_coc.py:148-149: alpha, beta are shared by all regions and heads of a module). */
extern "C" int vrnet_cluster_ab_reduce_multi(int n, const void* const* partial, const long* blocks, float* const* dalpha,
                                             float* const* dbeta, const int* accumulate, void* stream) {
  VR_CHECK_ARG(n > 0 && partial && blocks && dalpha && dbeta && accumulate, "cluster_ab_reduce_multi: bad arguments");
  if (vr_ablated("cluster")) return VR_OK;
  for (int i0 = 0; i0 < n; i0 += AB_MAX) {
    AbTable t{};
    const int m = n - i0 < AB_MAX ? n - i0 : AB_MAX;
    for (int i = 0; i < m; ++i) {
      VR_CHECK_ARG(partial[i0 + i] && dalpha[i0 + i] && dbeta[i0 + i] && blocks[i0 + i] > 0 && blocks[i0 + i] < (1L << 31),
                   "cluster_ab_reduce_multi: bad entry");
      t.partial[i] = reinterpret_cast<const float*>(partial[i0 + i]);
      t.dalpha[i] = dalpha[i0 + i];
      t.dbeta[i] = dbeta[i0 + i];
      t.blocks[i] = (int)blocks[i0 + i];
      t.accumulate[i] = accumulate[i0 + i] ? 1 : 0;
    }
    hipLaunchKernelGGL(cluster_ab_reduce_multi_kernel, dim3(m), dim3(256), 0, vr_stream(stream), t);
    VR_LAUNCH_CHECK("cluster_ab_reduce_multi");
  }
  return VR_OK;
}
