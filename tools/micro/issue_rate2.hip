// micro-benchmark, second sheet (see issue_rate.hip): the select / mask idioms of the physics kernel on one SIMD shared by 1–4 waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate2 tools/micro/issue_rate2.hip && /tmp/issue_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
enum { CND_VCC_SELF, CND_SGPR_SELF, CND_VCC_INDEP, CND_SGPR_INDEP, CND_VV_VCC, VAND, VBFI, EXECMOV, VMOV, FMA_SGPR, READLANE_USE, FMAC, ADDU32, WRITELANE, SMOV64LIT, SNOP0, SNOP1, WAITCNT0,
       DSREAD2, DSWRITE, ADDC, CND_AFTER_CMP, MULMASK, N_OPS };
static const char *NAMES[] = {"v_cndmask e32 v,0,v,vcc (self)", "v_cndmask e64 v,0,v,s[] (self)", "v_cndmask e32 d,0,v,vcc (indep)", "v_cndmask e64 d,0,v,s[] (indep)", "v_cndmask e32 v,v,v,vcc",
                              "v_and_b32 v,vmask,v", "v_bfi_b32 v,vmask,v,v", "s_mov exec,m; v_mov; s_mov exec,-1", "v_mov_b32", "v_fma_f32 v,s,v,v", "v_readlane; s_nop 1; v_fmac v,s,v",
                              "v_fmac_f32 (VOP2)", "v_add_u32", "v_writelane_b32", "s_mov_b64 s,lit64", "s_nop 0", "s_nop 1", "s_waitcnt lgkmcnt(0) (idle)", "ds_read2_b32", "ds_write_b32",
                              "v_addc_co_u32 vcc", "v_cmp_lt_f32 vcc; s_nop 1; v_cndmask", "v_mul_f32 v,vmask01,v"};
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
  asm volatile("s_mov_b64 vcc, %0" ::"s"(sm) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if constexpr (OP == CND_VCC_SELF) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == CND_SGPR_SELF) {
#define X(j) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[j]) : "s"(sm));
        REP16(X)
#undef X
      } else if constexpr (OP == CND_VCC_INDEP) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(p[j]) : "v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == CND_SGPR_INDEP) {
#define X(j) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(p[j]) : "v"(a[j]), "s"(sm));
        REP16(X)
#undef X
      } else if constexpr (OP == CND_VV_VCC) {
#define X(j) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(p[j]) : "v"(a[j]), "v"(q[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == VAND) {
#define X(j) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[j]) : "v"(vmask));
        REP16(X)
#undef X
      } else if constexpr (OP == VBFI) {
#define X(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[j]) : "v"(vmask), "v"(q[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == EXECMOV) {
#define X(j) asm volatile("s_mov_b64 exec, %1\n v_mov_b32 %0, %2\n s_mov_b64 exec, -1" : "+v"(a[j]) : "s"(sm), "v"(q[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == VMOV) {
#define X(j) asm volatile("v_mov_b32 %0, %1" : "=v"(p[j]) : "v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == FMA_SGPR) {
#define X(j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "s"(sb), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == READLANE_USE) {
#define X(j) asm volatile("v_readlane_b32 %1, %2, 5\n s_nop 1\n v_fmac_f32 %0, %1, %3" : "+v"(a[j]), "=s"(s[j]) : "v"(q[j]), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == FMAC) {
#define X(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == ADDU32) {
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(addr));
        REP16(X)
#undef X
      } else if constexpr (OP == WRITELANE) {
#define X(j) asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(a[j]) : "s"(s[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == SMOV64LIT) {
#define X(j) asm volatile("s_mov_b64 %0, 0x12345678" : "=s"(sm));
        REP16(X)
#undef X
      } else if constexpr (OP == SNOP0) {
#define X(j) asm volatile("s_nop 0");
        REP16(X)
#undef X
      } else if constexpr (OP == SNOP1) {
#define X(j) asm volatile("s_nop 1");
        REP16(X)
#undef X
      } else if constexpr (OP == WAITCNT0) {
#define X(j) asm volatile("s_waitcnt lgkmcnt(0)");
        REP16(X)
#undef X
      } else if constexpr (OP == DSREAD2) {
#define X(j) asm volatile("ds_read2_b32 %0, %1 offset0:" #j " offset1:" #j "+64" : "=v"(*(double *)&lds[0]) : "v"(addr));
        (void)0;
#undef X
#define X(j) { double dd; asm volatile("ds_read2_b32 %0, %1 offset0:" #j " offset1:" #j "+64" : "=v"(dd) : "v"(addr)); q[j] = __builtin_bit_cast(float, (int)__builtin_bit_cast(long long, dd)); }
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (OP == DSWRITE) {
#define X(j) asm volatile("ds_write_b32 %0, %1 offset:" #j "*256" ::"v"(addr), "v"(a[j]));
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (OP == ADDC) {
#define X(j) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[j]) : "v"(addr) : "vcc");
        REP16(X)
#undef X
      } else if constexpr (OP == CND_AFTER_CMP) {
#define X(j) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32_e32 %0, 0, %1, vcc" : "=v"(p[j]) : "v"(a[j]), "v"(b) : "vcc");
        REP16(X)
#undef X
      } else if constexpr (OP == MULMASK) {
#define X(j) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j]) : "v"(fmask));
        REP16(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
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
