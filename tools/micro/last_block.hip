// Micro-benchmark / stress test of the "last-arriving workgroup finishes the reduction" pattern on gfx950 (8 XCDs, one L2
// each).  A streaming kernel (reads x, writes y = 2x: dirty lines in every L2) leaves per-workgroup column partials; the
// cross-workgroup sum is done by
//   A  a second kernel (the library's form up to round 4),
//   B  the last workgroup to take a ticket, partials written / read with agent-scope RELAXED atomics (global_store / load
//      sc1: written through to / read from the device-coherent level) + s_waitcnt vmcnt(0) before the ticket -- no fence, so
//      no buffer_wbl2 (write-back of the whole L2) per workgroup,
//   C  the same with plain stores and __threadfence() (release / acquire fences: buffer_wbl2 sc1 + buffer_inv sc1).
// Checks B and C against A bit for bit on every iteration (beside a second stream that thrashes L2) and times the three.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: partials only (A), 1: sc1 atomics + ticket (B), 2: fences + ticket (C)
__global__ __launch_bounds__(256) void stream_kernel(const float* x, float* y, long rows, int C, long rows_per_wg, double* part,
                                                     unsigned* ticket, double* out) {
  const int tid = threadIdx.x, c = tid % C, ty = tid / C, RP = 256 / C;
  const long r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  double s = 0;
  for (long r = r0 + ty; r < r1; r += RP) {
    const float v = x[r * C + c];
    y[r * C + c] = 2.f * v;
    s += (double)v;
  }
  __shared__ double sm[256];
  __shared__ unsigned last;
  sm[tid] = s;
  __syncthreads();
  if (ty == 0) {
    for (int q = 1; q < RP; ++q) s += sm[q * C + c];
    double* p = part + (long)blockIdx.x * C + c;
    if (MODE == 1) __hip_atomic_store(p, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = s;
  }
  if (MODE == 0) return;
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else __threadfence();
  __syncthreads();
  if (tid == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (last != gridDim.x - 1) return;
  if (MODE == 2) __threadfence();
  // the last workgroup: sum over workgroups in index order, 256 / C lanes per column, lane order through LDS
  double t = 0;
  for (int b = ty; b < (int)gridDim.x; b += RP) {
    const double* p = part + (long)b * C + c;
    t += MODE == 1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  }
  __syncthreads();
  sm[tid] = t;
  __syncthreads();
  if (ty == 0) {
    for (int q = 1; q < RP; ++q) t += sm[q * C + c];
    out[c] = t;
  }
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void final_kernel(const double* part, int nwg, int C, double* out) {
  const int tid = threadIdx.x, c = tid % C, ty = tid / C, RP = 256 / C;
  __shared__ double sm[256];
  double t = 0;
  for (int b = ty; b < nwg; b += RP) t += part[(long)b * C + c];
  sm[tid] = t;
  __syncthreads();
  if (ty == 0) {
    for (int q = 1; q < RP; ++q) t += sm[q * C + c];
    out[c] = t;
  }
}

__global__ void thrash_kernel(float* z, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) z[i] = z[i] * 1.0001f + 1.f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 300;
  struct Shape { long rows; int C; int nwg; } shapes[] = {{131072, 64, 512}, {32768, 128, 256}, {8192, 64, 64}, {2048, 128, 32}, {131072, 64, 2048}};
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  float* z; const long zn = 64L << 20;
  CK(hipMalloc(&z, zn * 4)); CK(hipMemset(z, 0, zn * 4));
  for (auto sh : shapes) {
    const long n = sh.rows * sh.C;
    float *x, *y; double *part, *oa, *ob, *oc; unsigned* ticket;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&part, (long)sh.nwg * sh.C * 8));
    CK(hipMalloc(&oa, sh.C * 8)); CK(hipMalloc(&ob, sh.C * 8)); CK(hipMalloc(&oc, sh.C * 8)); CK(hipMalloc(&ticket, 4));
    CK(hipMemset(ticket, 0, 4));
    std::vector<float> hx(n);
    const long rpw = (sh.rows + sh.nwg - 1) / sh.nwg;
    std::vector<double> ha(sh.C), hb(sh.C), hc(sh.C);
    int bad_b = 0, bad_c = 0;
    for (int it = 0; it < iters; ++it) {
      if (it % 50 == 0) {       // new data now and then
        for (long i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u + it * 40503u) % 10007) / 10007.f - 0.3f;
        CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
      }
      thrash_kernel<<<512, 256, 0, s2>>>(z, zn);
      stream_kernel<0><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, oa);
      final_kernel<<<1, 256, 0, s1>>>(part, sh.nwg, sh.C, oa);
      CK(hipMemsetAsync(part, 0xff, (long)sh.nwg * sh.C * 8, s1));      // stale partials must not survive
      stream_kernel<1><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, ob);
      CK(hipMemsetAsync(part, 0xff, (long)sh.nwg * sh.C * 8, s1));
      stream_kernel<2><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, oc);
      CK(hipMemcpyAsync(ha.data(), oa, sh.C * 8, hipMemcpyDeviceToHost, s1));
      CK(hipMemcpyAsync(hb.data(), ob, sh.C * 8, hipMemcpyDeviceToHost, s1));
      CK(hipMemcpyAsync(hc.data(), oc, sh.C * 8, hipMemcpyDeviceToHost, s1));
      CK(hipStreamSynchronize(s1));
      if (memcmp(ha.data(), hb.data(), sh.C * 8)) ++bad_b;
      if (memcmp(ha.data(), hc.data(), sh.C * 8)) ++bad_c;
    }
    CK(hipDeviceSynchronize());
    // timing, alone on the device: 200 back-to-back launches each
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[4];
    for (int v = 0; v < 4; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, s1));
        for (int i = 0; i < 200; ++i) {
          if (v == 0) { stream_kernel<0><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, oa); final_kernel<<<1, 256, 0, s1>>>(part, sh.nwg, sh.C, oa); }
          if (v == 1) stream_kernel<1><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, ob);
          if (v == 2) stream_kernel<2><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, oc);
          if (v == 3) stream_kernel<0><<<sh.nwg, 256, 0, s1>>>(x, y, sh.rows, sh.C, rpw, part, ticket, oa);
        }
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[v], e0, e1));
      }
    }
    printf("rows %7ld C %3d wgs %4d | mismatches vs two-kernel form: sc1-atomics %d, fences %d of %d | us per launch: two kernels %.2f, "
           "last-WG sc1 %.2f, last-WG fences %.2f, partials only %.2f\n", sh.rows, sh.C, sh.nwg, bad_b, bad_c, iters,
           ms[0] * 5, ms[1] * 5, ms[2] * 5, ms[3] * 5);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(part)); CK(hipFree(oa)); CK(hipFree(ob)); CK(hipFree(oc)); CK(hipFree(ticket));
  }
  return 0;
}
