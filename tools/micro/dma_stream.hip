// How fast can workgroups stream a row-major [M][K] fp32 matrix into LDS by global_load_lds_dwordx4, as the igemm
// loaders do?  Each workgroup owns ROWS consecutive rows and walks K in chunks of CH bytes per row per stage
// (stage = ROWS x CH bytes = 8 KB), NST-1 stages in flight.  Reports TB/s for K = 64..2560 floats.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, int NST>
__global__ __launch_bounds__(256) void k(const float* a, int M, int K, float* sink) {
  constexpr int ROWS = 8192 / CH;            // rows per stage
  constexpr int LPR = CH / 16;               // lanes (16 B each) per row chunk
  __shared__ __attribute__((aligned(16))) float sm[NST][2048];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long row0 = (long)blockIdx.x * ROWS;
  const int nst = (K * 4) / CH;
  int r[2], q[2];
  for (int i = 0; i < 2; ++i) { const int sl = (wave * 2 + i) * 64 + lane; r[i] = sl / LPR; q[i] = sl % LPR; }
  auto issue = [&](int s, int buf) {
    for (int i = 0; i < 2; ++i) {
      const float* src = a + (row0 + r[i]) * K + (long)s * (CH / 4) + 4 * q[i];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(&sm[buf][(wave * 2 + i) * 256]), 16, 0, 0);
    }
  };
  for (int s = 0; s < NST - 1 && s < nst; ++s) issue(s, s);
  float acc = 0.f;
  for (int s = 0; s < nst; ++s) {
    if (s + NST - 2 < nst - 0 && NST == 3 && s + 1 < nst) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NST - 1 < nst) issue(s + NST - 1, (s + NST - 1) % NST);
    acc += sm[s % NST][tid * 4];
  }
  if (acc == 123.456f) sink[0] = acc;
}
template <int CH, int NST>
void run(const float* a, int M, int K, float* sink) {
  constexpr int ROWS = 8192 / CH;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, NST>), dim3(M / ROWS), dim3(256), 0, 0, a, M, K, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("  chunk %4d B x %3d rows, %d stages: %7.3f ms  %6.2f TB/s\n", CH, ROWS, NST, best, (double)M * K * 4 / best / 1e9);
}
int main() {
  const int M = 131072;
  float *a, *sink; hipMalloc(&a, (size_t)M * 2560 * 4); hipMalloc(&sink, 64);
  hipMemset(a, 0, (size_t)M * 2560 * 4);
  for (int K : {512, 2560, 128}) {
    printf("M=%d K=%d (%.0f MB)\n", M, K, (double)M * K * 4 / 1e6);
    run<128, 3>(a, M, K, sink); run<256, 3>(a, M, K, sink); run<512, 3>(a, M, K, sink);
    if (K >= 256) { run<1024, 3>(a, M, K, sink); }
    run<128, 2>(a, M, K, sink);
  }
  return 0;
}
