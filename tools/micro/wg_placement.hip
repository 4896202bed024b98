// Where does the dispatcher put the workgroups of an under-filled grid?  Each workgroup (256 threads, LDS bytes given on the
// command line, spinning ~30 us so that the whole grid is resident together) records the XCC / SE / CU it runs on; the host
// prints how many CUs were used and the histogram of workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 wg_placement.hip -o wg_placement.bin ;  ./wg_placement.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, long spin) {
  extern __shared__ float sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long t0 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  while (__builtin_amdgcn_s_memtime() - t0 < spin) acc += sm[threadIdx.x & 63];
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (acc == 1.2345f) out[0] = 0;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 8192 * 8);
  for (int lds : {43008, 57344, 86016}) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int grid : {256, 320, 512, 640}) {
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d, 3000L);   // ~30 us at 100 MHz
      (void)hipDeviceSynchronize();
      std::vector<unsigned> h(2 * grid);
      (void)hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per_cu;
      for (int i = 0; i < grid; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
      }
      int hist[16] = {0};
      for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
      printf("LDS %6d B/WG (%d fit per CU)  grid %4d: %3zu CUs used; CUs with 1/2/3/4 workgroups: %d / %d / %d / %d\n", lds, 163840 / lds,
             grid, per_cu.size(), hist[1], hist[2], hist[3], hist[4]);
    }
  }
  return 0;
}
