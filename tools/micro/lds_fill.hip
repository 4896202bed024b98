// Per-CU rate of bringing L2-resident data into LDS: (0) global_load_lds_dwordx4 (LDS-DMA, what the ring kernels use),
// (1) global_load_dwordx4 to registers + ds_write_b128, (2) global_load_dwordx4 to registers only.  Every workgroup walks
// its own 64 KB window over and over (L2 hits after the first pass); NP 1 KB pieces per wave in flight per round.
//   hipcc --offload-arch=gfx950 -O3 lds_fill.hip -o lds_fill.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NP>
__global__ __launch_bounds__(256) void k(const float* src, int iters, float* sink, long win_floats) {
  __shared__ __attribute__((aligned(16))) float sm[4 * NP * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* base = src + (long)blockIdx.x * win_floats;
  float acc = 0.f;
  int off = 0;
  for (int it = 0; it < iters; ++it) {
    f32x4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float* p = base + off + ((wave * NP + i) * 64 + lane) * 4;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(&sm[(wave * NP + i) * 256]), 16, 0, 0);
      } else {
        v[i] = *reinterpret_cast<const f32x4*>(p);
      }
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) *reinterpret_cast<f32x4*>(&sm[((wave * NP + i) * 64 + lane) * 4]) = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) acc += v[i][0];
    }
    off += 4 * NP * 256;
    if (off + 4 * NP * 256 > win_floats) off = 0;
    asm volatile("" ::: "memory");
  }
  if (MODE != 2) { __syncthreads(); acc = sm[tid]; }
  if (acc == 123.456f) sink[0] = acc;
}
template <int MODE, int NP>
void run(const float* a, float* sink, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NP>), dim3(blocks), dim3(256), 0, 0, a, iters, sink, 16384L);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = (double)blocks * iters * 4 * NP * 1024;
  printf("mode %d (%s) %d pieces/wave/round, %d WG/CU: %7.3f ms  %6.2f TB/s  %5.1f B/cycle/CU @2.4GHz\n", MODE,
         MODE == 0 ? "LDS-DMA x4      " : (MODE == 1 ? "load x4 + ds_write" : "load x4 only    "), NP, blocks / 256, best,
         bytes / best / 1e9, bytes / 256 / (best * 1e-3 * 2.4e9));
}
int main() {
  float *a, *sink; (void)hipMalloc(&a, (size_t)1024 * 65536); (void)hipMalloc(&sink, 64);
  (void)hipMemset(a, 0, (size_t)1024 * 65536);
  for (int b = 256; b <= 1024; b *= 2) {
    run<0, 2>(a, sink, b); run<0, 4>(a, sink, b); run<0, 8>(a, sink, b);
    run<1, 2>(a, sink, b); run<1, 4>(a, sink, b); run<1, 8>(a, sink, b);
    run<2, 4>(a, sink, b); run<2, 8>(a, sink, b);
  }
  return 0;
}
