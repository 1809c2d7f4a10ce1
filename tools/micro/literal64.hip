// tools/micro/literal64.hip — what does gfx950 do with a 32-bit literal in s_mov_b64?
//   hipcc --offload-arch=gfx950 -O3 -o literal64 tools/micro/literal64.hip && ./literal64
// hipcc (LLVM 22, ROCm 7.2) materialises the 64-bit constant 0xfffffffffffc0000 as `s_mov_b64 s[0:1], 0xfffffffffffc0000`, i.e. ONE 32-bit
// literal 0xfffc0000 that it expects the hardware to sign-extend.  The kernel reads back what the SGPR pair holds, and what a lane mask
// made from that constant (llvm.amdgcn.inverse.ballot) selects in lanes 0-63.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned long long *out) {
  unsigned long long a, b;
  asm volatile("s_mov_b64 %0, 0xfffc0000" : "=s"(a));                 // the encoding in question, written by hand
  { unsigned lo, hi; asm volatile("s_mov_b32 %0, 0xfffc0000\n s_mov_b32 %1, -1" : "=s"(lo), "=s"(hi)); b = ((unsigned long long)hi << 32) | lo; }
  bool sel = __builtin_amdgcn_inverse_ballot_w64(0xfffffffffffc0000ull);   // compiler's choice of materialisation
  unsigned long long got = __builtin_amdgcn_ballot_w64(sel);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = got; }
}

int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("s_mov_b64 s[..], 0xfffc0000          -> %016llx\n", h[0]);
  printf("two s_mov_b32 (0xfffc0000, -1)       -> %016llx\n", h[1]);
  printf("inverse_ballot(0xfffffffffffc0000)   -> lanes %016llx   (%s)\n", h[2], h[2] == 0xfffffffffffc0000ull ? "as written" : "UPPER LANES LOST");
  return 0;
}
