// How many 64-lane workgroups with a given dynamic LDS size (and VGPR budget) are resident on one CU of gfx950?  Each workgroup finds its CU
// (HW_ID + XCC_ID), raises that CU's counter, records the maximum it saw, spins, lowers the counter.
// build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/lds_occupancy tools/micro/lds_occupancy.hip ; run: gpurun_out/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_occ(int *cnt, int *mx, int spin_us) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  const unsigned key = ((xcc & 0xf) << 8) | (se << 5) | (sh << 4) | cu;
  if (threadIdx.x == 0) {
    lds[0] = 1.f;
    int now = atomicAdd(&cnt[key], 1) + 1;
    atomicMax(&mx[key], now);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
    atomicSub(&cnt[key], 1);
  }
}

int main() {
  int *cnt, *mx;
  hipMalloc(&cnt, 4096 * sizeof(int));
  hipMalloc(&mx, 4096 * sizeof(int));
  hipFuncSetAttribute((const void *)k_occ, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int sizes[] = {16384, 16128, 15872, 15616, 15360, 15104, 14848, 14592, 14336, 14080, 13824, 13312, 12288, 10240, 8192};
  for (int lds : sizes) {
    hipMemset(cnt, 0, 4096 * sizeof(int));
    hipMemset(mx, 0, 4096 * sizeof(int));
    hipLaunchKernelGGL(k_occ, dim3(256 * 24), dim3(64), lds, 0, cnt, mx, 300);
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), mx, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    int best = 0, cus = 0, lo = 1 << 30;
    for (int v : h) if (v > 0) { cus++; if (v > best) best = v; if (v < lo) lo = v; }
    printf("lds %6d B: CUs seen %3d, max resident workgroups per CU %2d (min over CUs %2d)\n", lds, cus, best, lo);
  }
  return 0;
}
