// micro-benchmark, third sheet (see issue_rate.hip): what makes v_cndmask_b32 slow of the physics kernel on one SIMD shared by 1–4 waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate3 tools/micro/issue_rate3.hip && /tmp/issue_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
enum { SMOVVCC_CND, CMP_4CND, CND64_VCC, CND32_3FMA, SMOV_CND64, CMP64_CND64, CND32_REWRITE4, CND32_ALT_E64, CND32_NOP, N_OPS };
static const char *NAMES[] = {"s_mov vcc,s; v_cndmask e32 vcc", "v_cmp vcc; s_nop 1; 4x cndmask e32", "v_cndmask e64 v,0,v,vcc", "cndmask e32 vcc + 3 v_fma", "s_mov s[],lit; v_cndmask e64 s[]",
                              "v_cmp_e64 s[]; s_nop 1; cndmask e64 s[]", "s_mov vcc; 4x cndmask e32", "cndmask e32 vcc, cndmask e64 s[] alternating", "cndmask e32 vcc; s_nop 0"};
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float b, float c, unsigned long long mask) {
  extern __shared__ float lds[];
  float a[16], p[16], q[16];
  unsigned s[16];
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x + i; s[i] = i; p[i] = i; q[i] = 2 * i; }
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 4;
  const unsigned vmask = (mask >> (threadIdx.x & 63)) & 1 ? 0xffffffffu : 0u;
  const float fmask = (mask >> (threadIdx.x & 63)) & 1 ? 1.f : 0.f;
  unsigned long long sm = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
  float sb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
  float acc0 = 0;
  asm volatile("s_mov_b64 vcc, %0" ::"s"(sm) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long sm2 = sm;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if constexpr (OP == SMOVVCC_CND) {
#define X(j) asm volatile("s_mov_b64 vcc, %1\n v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(a[j]) : "s"(sm) : "vcc");
        REP16(X)
#undef X
      } else if constexpr (OP == CMP_4CND) {
#define X(j) asm volatile("v_cmp_lt_f32 vcc, %4, %5\n s_nop 1\n v_cndmask_b32_e32 %0, 0, %0, vcc\n v_cndmask_b32_e32 %1, 0, %1, vcc\n v_cndmask_b32_e32 %2, 0, %2, vcc\n v_cndmask_b32_e32 %3, 0, %3, vcc" : "+v"(a[j]), "+v"(p[j]), "+v"(q[j]), "+v"(a[(j + 8) & 15]) : "v"(b), "v"(c) : "vcc");
        REP16(X)
#undef X
      } else if constexpr (OP == CND64_VCC) {
#define X(j) asm volatile("v_cndmask_b32_e64 %0, 0, %0, vcc" : "+v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == CND32_3FMA) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc\n v_fma_f32 %1, %3, %4, %1\n v_fma_f32 %2, %3, %4, %2\n v_fma_f32 %1, %3, %4, %1" : "+v"(a[j]), "+v"(p[j]), "+v"(q[j]) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == SMOV_CND64) {
#define X(j) asm volatile("s_mov_b64 %1, 0x0f0f3355\n v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[j]), "=s"(sm2));
        REP16(X)
#undef X
      } else if constexpr (OP == CMP64_CND64) {
#define X(j) asm volatile("v_cmp_lt_f32_e64 %1, %2, %3\n s_nop 1\n v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[j]), "=s"(sm2) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == CND32_REWRITE4) {
#define X(j) asm volatile("s_mov_b64 vcc, %4\n v_cndmask_b32_e32 %0, 0, %0, vcc\n v_cndmask_b32_e32 %1, 0, %1, vcc\n v_cndmask_b32_e32 %2, 0, %2, vcc\n v_cndmask_b32_e32 %3, 0, %3, vcc" : "+v"(a[j]), "+v"(p[j]), "+v"(q[j]), "+v"(a[(j + 8) & 15]) : "s"(sm) : "vcc");
        REP16(X)
#undef X
      } else if constexpr (OP == CND32_ALT_E64) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc\n v_cndmask_b32_e64 %1, 0, %1, %2" : "+v"(a[j]), "+v"(p[j]) : "s"(sm));
        REP16(X)
#undef X
      } else if constexpr (OP == CND32_NOP) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc\n s_nop 0" : "+v"(a[j]));
        REP16(X)
#undef X
      }
    }
  }
  acc0 += (float)sm2;
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = acc0;
  for (int i = 0; i < 16; i++) acc += a[i] + (float)s[i] + p[i] + q[i];
  acc += (float)sm;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(float *out, unsigned long long *cyc) {
  const int iters = 4000, nwg = 256;
  hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-38s", NAMES[OP]);
  for (int kw = 1; kw <= 4; kw++) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(nwg), dim3(256 * kw), 96 * 1024, 0, out, cyc, iters, 1.0001f, 0.5f, 0x00ff00ff0f0f3355ull);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(nwg * 4 * kw);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double n = (double)iters * 32;
    printf(" | k=%d %6.2f/wave %5.2f/SIMD %.2f GHz", kw, mean / n, mean / n / kw, mean / (ms * 1e6));
  }
  printf("\n");
}
template <int OP> void all(float *out, unsigned long long *cyc) { if constexpr (OP < N_OPS) { run<OP>(out, cyc); all<OP + 1>(out, cyc); } }
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  printf("s_memtime ticks per instruction (or per listed group), as one wave sees them and per SIMD; GHz = ticks / wall time\n");
  all<0>(out, cyc);
  return 0;
}
