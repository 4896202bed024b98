// Issue rate of v_mfma_f32_32x32x16_bf16 as a function of accumulator-chain structure: NCH independent accumulators,
// 24 MFMAs per iteration, chain-major (all of chain 0, then chain 1, ...) or interleaved (round robin).
//   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH, bool INTER>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NCH];
  for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.f - threadIdx.x * 1e-3f + e); }
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(a), "+v"(b));
    if (INTER) {
#pragma unroll
      for (int j = 0; j < 24 / NCH; ++j)
#pragma unroll
        for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 24 / NCH; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NCH; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH, bool INTER>
void run(int blocks) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCH, INTER>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double mfmas_per_simd = (double)blocks / 256.0 * iters * 24;
  printf("chains %d %-11s %d waves/SIMD: %7.3f ms  %6.1f ns-cycles@2.4GHz per MFMA per SIMD, %6.1f TF bf16\n", NCH,
         INTER ? "interleaved" : "chain-major", blocks / 256, ms, ms * 1e-3 * 2.4e9 / mfmas_per_simd,
         (double)blocks * 4 * iters * 24 * 32768.0 / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int b = 256; b <= 512; b += 256) {
    run<1, false>(b); run<2, false>(b); run<2, true>(b); run<4, false>(b); run<4, true>(b); run<8, true>(b);
  }
  return 0;
}
