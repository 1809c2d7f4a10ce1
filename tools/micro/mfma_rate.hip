// micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2_f32 with 20 / 5 independent accumulators, one wave per SIMD,
// 256 workgroups of 256 threads; prints cycles per MFMA per SIMD (s_memtime at 100 MHz -> shader clocks estimated from wall time)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(16))) f16v;
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = f4{0, 0, 0, 0};
  float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
  float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 256 * 256 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg : {256, 512}) {
    for (int rep = 0; rep < 2; rep++) {
      int iters = 2000;
      hipEventRecord(e0); hipLaunchKernelGGL(k16<20>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double n = (double)iters * 20 * (wg / 256.0);
      printf("16x16x4 f32, %d WGs: %.3f ms, %.1f ns per MFMA per SIMD-slot, %.1f TFLOP/s\n", wg, ms, ms * 1e6 / n, 2048.0 * iters * 20 * wg * 4 / (ms * 1e-3) / 1e12);
      hipEventRecord(e0); hipLaunchKernelGGL(k32<5>, dim3(wg), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("32x32x2 f32, %d WGs: %.3f ms, %.1f TFLOP/s\n", wg, ms, 4096.0 * iters * 5 * wg * 4 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
