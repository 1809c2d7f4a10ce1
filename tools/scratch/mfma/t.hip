#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* in) {
  int lane = threadIdx.x;
  float x0 = in[lane], x1 = in[lane + 64], x2 = in[lane + 128], x3 = in[lane + 192];
  auto s01 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1), false, false);
  auto s23 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x2), __builtin_bit_cast(unsigned, x3), false, false);
  auto p = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
  float a = __builtin_bit_cast(float, p[0]);
  out[lane + 256] = a;
  // A[i][kk] = (i+1) if kk==0 else 0 ; B[kk][j] = (j+1)*10 if kk==0 : expect C[i][j] = (i+1)*(j+1)*10
  float A = (lane / 16 == 0) ? (float)(lane % 16 + 1) : 0.f, B = (lane / 16 == 0) ? (float)((lane % 16 + 1) * 10) : 0.f;
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, c, 0, 0, 0);
  out[lane] = c[0]; out[lane + 64] = c[1]; out[lane + 128] = c[2]; out[lane + 192] = c[3];
}
int main() {
  float h[320], *d, *o;
  for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) h[r * 64 + l] = 1000.f * r + l;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("packed a: "); for (int l = 0; l < 64; l += 1) printf("%g ", h[256 + l]); printf("\n");
  for (int v = 0; v < 4; v++) { printf("c[%d]: ", v); for (int l = 0; l < 64; l += 1) printf("%g ", h[v * 64 + l]); printf("\n"); }
  return 0;
}
