// micro-benchmark: what ONE SIMD of an MI355X sustains per instruction class when 1, 2, 3 or 4 waves share it — the question behind the physics kernel's
// "three waves per SIMD, the fourth buys nothing" (DESIGN.md §4).  One workgroup of 256·k threads per CU (96 KB of LDS keep a second one away), so k waves sit on
// every SIMD; each wave runs `iters` × 32 independent instructions of one class (16 chains: no dependency stall) between two s_memtime reads.
// Prints cycles per instruction as ONE wave sees them and per SIMD (÷ k): a class whose per-SIMD figure stops falling at k = 2 has its pipe saturated by two waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/micro/issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { FMA, DPP, CNDMASK, READLANE, FMA64, PKFMA, RCP, LDSREAD, BPERMUTE, CMP, SALU, IMAD, MULLO, FMA_SALU, FMA_LDS, PERMLANE, READFIRST, NOPS };
static const char *NAMES[] = {"v_fma_f32", "v_add_f32 dpp row_shr", "v_cndmask_b32", "v_readlane_b32", "v_fma_f64", "v_pk_fma_f32", "v_rcp_f32",
                              "ds_read_b32", "ds_bpermute_b32", "v_cmp_lt_f32 (sgpr pair)", "s_add_u32", "v_mad_u32_u24", "v_mul_lo_u32",
                              "v_fma_f32 + s_add_u32 pairs", "v_fma_f32 + ds_read_b32 pairs", "v_permlane32_swap", "v_readfirstlane_b32", "-"};

template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *cyc, int iters, float b, float c) {
  extern __shared__ float lds[];
  float a[16];
  double d[8];
  float p[16][2];
  unsigned s[16];
  for (int i = 0; i < 16; i++) { a[i] = threadIdx.x + i; s[i] = i; p[i][0] = i; p[i][1] = -i; }
  for (int i = 0; i < 8; i++) d[i] = threadIdx.x * 0.5 + i;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 4;
  typedef float __attribute__((ext_vector_type(2))) f2;
  f2 pk[16]; for (int i = 0; i < 16; i++) pk[i] = f2{(float)i, (float)-i};
  f2 pb = f2{b, b}, pc = f2{c, c};
  const double db = b, dc = c;
  unsigned long long m0 = 0; unsigned sacc = 0;
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(b), "v"(a[0]) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if constexpr (OP == FMA) {
#define X(j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
        REP16(X)
#undef X
      } else if constexpr (OP == DPP) {
#define X(j) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(b));
        REP16(X)
#undef X
      } else if constexpr (OP == CNDMASK) {
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : );
        REP16(X)
#undef X
      } else if constexpr (OP == READLANE) {
#define X(j) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[j]) : "v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == READFIRST) {
#define X(j) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s[j]) : "v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == FMA64) {
#define X(j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[j & 7]) : "v"(db), "v"(dc));
        REP16(X)
#undef X
      } else if constexpr (OP == PKFMA) {
#define X(j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk[j]) : "v"(pb), "v"(pc));
        REP16(X)
#undef X
      } else if constexpr (OP == RCP) {
#define X(j) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
        REP16(X)
#undef X
      } else if constexpr (OP == LDSREAD) {
#define X(j) asm volatile("ds_read_b32 %0, %1 offset:" #j "*256" : "=v"(a[j]) : "v"(addr));
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (OP == BPERMUTE) {
#define X(j) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[j]) : "v"(addr));
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (OP == CMP) {
#define X(j) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m0) : "v"(a[j]), "v"(b));
        REP16(X)
#undef X
      } else if constexpr (OP == SALU) {
#define X(j) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[j]) : : "scc");
        REP16(X)
#undef X
      } else if constexpr (OP == IMAD) {
#define X(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[j]) : "v"(addr));
        REP16(X)
#undef X
      } else if constexpr (OP == MULLO) {
#define X(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(addr));
        REP16(X)
#undef X
      } else if constexpr (OP == FMA_SALU) {
#define X(j) asm volatile("v_fma_f32 %0, %2, %3, %0\n s_add_u32 %1, %1, 3" : "+v"(a[j]), "+s"(s[j]) : "v"(b), "v"(c) : "scc");
        REP16(X)
#undef X
      } else if constexpr (OP == FMA_LDS) {
#define X(j) asm volatile("v_fma_f32 %0, %2, %3, %0\n ds_read_b32 %1, %4 offset:" #j "*256" : "+v"(a[j]), "=v"(p[j][0]) : "v"(b), "v"(c), "v"(addr));
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if constexpr (OP == PERMLANE) {
#define X(j) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[j]), "+v"(p[j][0]));
        REP16(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0;
  for (int i = 0; i < 16; i++) acc += a[i] + (float)s[i] + p[i][0] + p[i][1] + pk[i][0] + pk[i][1];
  for (int i = 0; i < 8; i++) acc += (float)d[i];
  acc += (float)m0 + sacc;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(float *out, unsigned long long *cyc, int per_iter_mult) {
  const int iters = 4000, nwg = 256;
  hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-30s", NAMES[OP]);
  for (int kw = 1; kw <= 4; kw++) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(nwg), dim3(256 * kw), 96 * 1024, 0, out, cyc, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(nwg * 4 * kw);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double n = (double)iters * 32 * per_iter_mult;
    printf(" | k=%d: %6.2f cyc/inst/wave %5.2f /SIMD (%.0f ns/k-inst)", kw, mean / n, mean / n / kw, ms * 1e6 / n * 1000.0);
  }
  printf("\n");
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  printf("cycles = s_memtime ticks; per instruction (pairs: per pair)\n");
  run<FMA>(out, cyc, 1); run<DPP>(out, cyc, 1); run<CNDMASK>(out, cyc, 1); run<READLANE>(out, cyc, 1); run<READFIRST>(out, cyc, 1); run<FMA64>(out, cyc, 1);
  run<PKFMA>(out, cyc, 1); run<RCP>(out, cyc, 1); run<LDSREAD>(out, cyc, 1); run<BPERMUTE>(out, cyc, 1); run<CMP>(out, cyc, 1); run<SALU>(out, cyc, 1);
  run<IMAD>(out, cyc, 1); run<MULLO>(out, cyc, 1); run<FMA_SALU>(out, cyc, 1); run<FMA_LDS>(out, cyc, 1); run<PERMLANE>(out, cyc, 1);
  return 0;
}
