// Multi-tensor parameter updates: one launch updates every parameter of the model (887 state_dict entries, 595
// optimised tensors at phi = l) instead of 3-5 elementwise launches per tensor.
// Replaces, for the training step that consumes the hot path's gradients:
//   torch.optim.SGD(momentum, nesterov=True) / torch.optim.Adam over the three parameter groups of train.py:460-473
//   ModelEMA.update (nets/yolo_training.py:465-475): ema = d * ema + (1 - d) * model over every floating tensor.
// Work is cut into fixed chunks of `chunk_elems` elements; (chunk_tensor[c], chunk_index[c]) name the tensor and
// the chunk inside it, so large and tiny tensors share one grid.  Pure HBM streams (SGD: 3 reads + 2 writes per
// element), arithmetic in fp32 in exactly torch's operation order.
#include "common.h"

namespace {

struct MtArgs {
  const long long* addrs;   // [K][n] device addresses
  const long* sizes;        // [n]
  const int* chunk_tensor;  // [n_chunks]
  const int* chunk_index;   // [n_chunks]
  const float* wd;          // [n] per-tensor weight decay (NULL = 0)
  int n, chunk_elems;
};

__device__ __forceinline__ bool mt_range(const MtArgs& a, int& t, long& lo, long& hi) {
  t = a.chunk_tensor[blockIdx.x];
  lo = (long)a.chunk_index[blockIdx.x] * a.chunk_elems;
  hi = lo + a.chunk_elems;
  if (hi > a.sizes[t]) hi = a.sizes[t];
  return lo < hi;
}

// torch.optim.SGD (_single_tensor_sgd): d = g + wd*p; buf = first ? d : mu*buf + d; d = nesterov ? d + mu*buf : buf;
// p -= lr*d   (dampening 0, maximize False)
__global__ __launch_bounds__(256) void mt_sgd_kernel(MtArgs a, float lr, float mu, int nesterov, int first) {
  int t;
  long lo, hi;
  if (!mt_range(a, t, lo, hi)) return;
  float* p = reinterpret_cast<float*>(a.addrs[t]);
  const float* g = reinterpret_cast<const float*>(a.addrs[a.n + t]);
  float* buf = reinterpret_cast<float*>(a.addrs[2 * a.n + t]);
  const float wd = a.wd ? a.wd[t] : 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float pv = p[i];
    float d = g[i];
    if (wd != 0.f) d = d + wd * pv;
    float b = d;
    if (mu != 0.f) {
      b = first ? d : mu * buf[i] + d;
      buf[i] = b;
      d = nesterov ? d + mu * b : b;
    }
    p[i] = pv - lr * d;
  }
}

// torch.optim.Adam (_single_tensor_adam, amsgrad False): g' = g + wd*p; m = b1*m + (1-b1) g'; v = b2*v + (1-b2) g'^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),  bc1 = 1 - b1^step, bc2 = 1 - b2^step
__global__ __launch_bounds__(256) void mt_adam_kernel(MtArgs a, float lr, float b1, float b2, float eps, float bc1,
                                                      float bc2_sqrt) {
  int t;
  long lo, hi;
  if (!mt_range(a, t, lo, hi)) return;
  float* p = reinterpret_cast<float*>(a.addrs[t]);
  const float* g = reinterpret_cast<const float*>(a.addrs[a.n + t]);
  float* m = reinterpret_cast<float*>(a.addrs[2 * a.n + t]);
  float* v = reinterpret_cast<float*>(a.addrs[3 * a.n + t]);
  const float wd = a.wd ? a.wd[t] : 0.f;
  const float step_size = lr / bc1;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const float pv = p[i];
    float gv = g[i];
    if (wd != 0.f) gv = gv + wd * pv;
    const float mv = m[i] + (gv - m[i]) * (1.f - b1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - step_size * (mv / denom);
  }
}

// ModelEMA.update: v *= d; v += (1 - d) * model
__global__ __launch_bounds__(256) void mt_ema_kernel(MtArgs a, float d) {
  int t;
  long lo, hi;
  if (!mt_range(a, t, lo, hi)) return;
  float* e = reinterpret_cast<float*>(a.addrs[t]);
  const float* m = reinterpret_cast<const float*>(a.addrs[a.n + t]);
  const float om = 1.f - d;
  for (long i = lo + threadIdx.x; i < hi; i += 256) e[i] = e[i] * d + om * m[i];
}

// dst = src (multi-tensor copy): builds the concatenated fc1|fc_v weights of every Cluster in one launch
__global__ __launch_bounds__(256) void mt_copy_kernel(MtArgs a) {
  int t;
  long lo, hi;
  if (!mt_range(a, t, lo, hi)) return;
  float* d = reinterpret_cast<float*>(a.addrs[t]);
  const float* s = reinterpret_cast<const float*>(a.addrs[a.n + t]);
  for (long i = lo + threadIdx.x; i < hi; i += 256) d[i] = s[i];
}

int mt_check(const char* name, const long long* addrs, const long* sizes, const int* ct, const int* ci, int n, int nc,
             int ce) {
  VR_CHECK_ARG(addrs && sizes && ct && ci && n > 0 && nc >= 0 && ce >= 256 && ce % 256 == 0, "%s: bad tensor table", name);
  return VR_OK;
}

}  // namespace

extern "C" int vrnet_mt_sgd_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                const float* weight_decay, int n_tensors, int n_chunks, int chunk_elems, float lr,
                                float momentum, int nesterov, int first_step, void* stream) {
  int rc = mt_check("mt_sgd", addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems);
  if (rc) return rc;
  VR_CHECK_ARG(!nesterov || momentum > 0.f, "mt_sgd: nesterov needs momentum > 0");
  if (n_chunks == 0) return VR_OK;
  MtArgs a{addrs, sizes, chunk_tensor, chunk_index, weight_decay, n_tensors, chunk_elems};
  hipLaunchKernelGGL(mt_sgd_kernel, dim3(n_chunks), dim3(256), 0, vr_stream(stream), a, lr, momentum, nesterov, first_step);
  VR_LAUNCH_CHECK("mt_sgd");
  return VR_OK;
}

extern "C" int vrnet_mt_adam_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                 const float* weight_decay, int n_tensors, int n_chunks, int chunk_elems, float lr,
                                 float beta1, float beta2, float eps, int step, void* stream) {
  int rc = mt_check("mt_adam", addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems);
  if (rc) return rc;
  VR_CHECK_ARG(step >= 1, "mt_adam: step counts from 1");
  if (n_chunks == 0) return VR_OK;
  MtArgs a{addrs, sizes, chunk_tensor, chunk_index, weight_decay, n_tensors, chunk_elems};
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(mt_adam_kernel, dim3(n_chunks), dim3(256), 0, vr_stream(stream), a, lr, beta1, beta2, eps, (float)bc1,
                     (float)sqrt(bc2));
  VR_LAUNCH_CHECK("mt_adam");
  return VR_OK;
}

extern "C" int vrnet_mt_ema_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                int n_tensors, int n_chunks, int chunk_elems, float decay, void* stream) {
  int rc = mt_check("mt_ema", addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems);
  if (rc) return rc;
  if (n_chunks == 0) return VR_OK;
  MtArgs a{addrs, sizes, chunk_tensor, chunk_index, nullptr, n_tensors, chunk_elems};
  hipLaunchKernelGGL(mt_ema_kernel, dim3(n_chunks), dim3(256), 0, vr_stream(stream), a, decay);
  VR_LAUNCH_CHECK("mt_ema");
  return VR_OK;
}

extern "C" int vrnet_mt_copy_f32(const long long* addrs, const long* sizes, const int* chunk_tensor, const int* chunk_index,
                                 int n_tensors, int n_chunks, int chunk_elems, void* stream) {
  int rc = mt_check("mt_copy", addrs, sizes, chunk_tensor, chunk_index, n_tensors, n_chunks, chunk_elems);
  if (rc) return rc;
  if (n_chunks == 0) return VR_OK;
  MtArgs a{addrs, sizes, chunk_tensor, chunk_index, nullptr, n_tensors, chunk_elems};
  hipLaunchKernelGGL(mt_copy_kernel, dim3(n_chunks), dim3(256), 0, vr_stream(stream), a);
  VR_LAUNCH_CHECK("mt_copy");
  return VR_OK;
}
