// Practical fp32-MFMA ceiling on the box: register-resident v_mfma_f32_32x32x2_f32 loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    if (rep == 2) printf("%s blocks=%d nacc=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, NACC, ms, flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<1>(256, "1wave/SIMD"); run<1>(512, "2waves/SIMD"); run<1>(1024, "4waves/SIMD"); run<1>(1792, "7waves/SIMD");
  run<4>(256, "1wave/SIMD"); run<4>(512, "2waves/SIMD"); run<4>(1024, "4w/SIMD");
  return 0;
}
