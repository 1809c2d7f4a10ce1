#include <hip/hip_runtime.h>
#include <stdio.h>
#define TM_DEV __device__ __forceinline__
TM_DEV int tm_f2i(float f) { int i; __builtin_memcpy(&i, &f, 4); return i; }
TM_DEV float tm_i2f(int i) { float f; __builtin_memcpy(&f, &i, 4); return f; }
template <int CTRL, int ROW_MASK, int BANK_MASK>
TM_DEV float tmw_dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, tm_f2i(v), CTRL, ROW_MASK, BANK_MASK, false);
  return v + tm_i2f(moved);
}
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x];
  float a = v;
  a = tmw_dpp_add<0x111, 0xf, 0xf>(a);
  out[64 + threadIdx.x] = a;
  a = tmw_dpp_add<0x112, 0xf, 0xf>(a);
  out[128 + threadIdx.x] = a;
  a = tmw_dpp_add<0x114, 0xf, 0xe>(a);
  out[192 + threadIdx.x] = a;
  a = tmw_dpp_add<0x118, 0xf, 0xc>(a);
  out[256 + threadIdx.x] = a;
  a = tmw_dpp_add<0x142, 0xa, 0xf>(a);
  out[320 + threadIdx.x] = a;
  a = tmw_dpp_add<0x143, 0xc, 0xf>(a);
  out[384 + threadIdx.x] = a;
  float s = __builtin_amdgcn_readlane(a, 63);
  float b = v;
  for (int off = 32; off >= 1; off >>= 1) b += __shfl_xor(b, off);
  out[threadIdx.x] = s - b;
  if (threadIdx.x == 0) { out[448] = s; out[449] = b; }
}
int main() {
  float h[64], o[512]; float ref = 0;
  for (int i = 0; i < 64; i++) { h[i] = (float)(i + 1); ref += h[i]; }
  float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 2048);
  hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, 2048, hipMemcpyDeviceToHost);
  printf("ref %f dpp %f shfl %f\n", ref, o[448], o[449]);
  for (int st = 1; st <= 6; st++) { printf("stage %d:", st); for (int i = 0; i < 64; i++) printf(" %g", o[64 * st + i]); printf("\n"); }
  return 0;
}
