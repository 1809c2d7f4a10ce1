// micro-benchmark: how the dispatcher spreads ONE-wave workgroups over a CU's four SIMDs when LDS allows R of them per CU and registers up to four per SIMD
// (the physics kernel's situation at 14 envs per CU: __launch_bounds__(64, 4), 9 LDS granules).  Every wave records HW_ID (gfx9: wave 3:0, simd 5:4, cu 11:8,
// sh 12, se 15:13; plus XCC_ID on gfx94x) and spins ~200 us so that the whole grid is resident together; the host prints the histogram of waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/simd_fill tools/micro/simd_fill.hip && /tmp/simd_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <array>
__global__ __launch_bounds__(64, 4) void k(unsigned *out, int spin) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  if (lds[63 - threadIdx.x] < 0) out[0] = 0;
}
int main() {
  unsigned *out; hipMalloc(&out, 256 * 16 * 2 * 4);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int per_cu : {12, 14, 8, 16}) {
    const int lds = per_cu == 12 ? 13000 : per_cu == 14 ? 11276 : per_cu == 8 ? 20000 : 10000;   // bytes: 160 KB / lds -> 12, 14, 8, 16 per CU
    const int grid = 256 * per_cu;
    for (int phase = 0; phase < 2; phase++) {
      // phase 0: one launch that fills the chip; phase 1: 3 x that many workgroups with random-ish run times: slots are refilled as waves retire
      hipMemset(out, 0xff, 256 * 16 * 2 * 4);
      if (phase == 0) hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, 0, out, 400000);
      else hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, 0, out, 400000);
      hipDeviceSynchronize();
      std::vector<unsigned> h(grid * 2);
      hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, std::array<int, 4>> per;   // (xcc, se, sh, cu) -> waves per simd
      for (int i = 0; i < grid; i++) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[(xcc << 12) | (se << 8) | (sh << 4) | cu][simd]++;
      }
      std::map<std::array<int, 4>, int> hist;
      for (auto &kv : per) { auto a = kv.second; hist[a]++; }
      printf("LDS %d B per workgroup (%d per CU), %d workgroups, pass %d: %zu CUs seen; (waves on simd0, 1, 2, 3) x CUs:", lds, per_cu, grid, phase, per.size());
      for (auto &kv : hist) printf("  (%d,%d,%d,%d) x %d", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
      printf("\n");
    }
  }
  return 0;
}
