// Where does v_mfma_f64_16x16x4_f64 put D[i][j]?  (csrc/wave_physics.h: TmwSchur assumes component i / 4, lane 16 (i % 4) + j — the layout
// CK's mfma_type<mfma_f64_16x16x4f64> describes: group_size 1, four groups per block.)  A = one-hot rows, B = one-hot columns, one product each.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o tools/micro/mfma_f64_layout tools/micro/mfma_f64_layout.hip && tools/micro/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out) {
  const int lane = threadIdx.x;
  // A[i][k]: lane 16 k + i; B[k][j]: lane 16 k + j.  A = (i + 1) at k = 0, B = (j + 1) * 100 at k = 0  =>  D[i][j] = (i + 1) (j + 1) 100
  double a = lane < 16 ? (double)(lane + 1) : 0.0, b = lane < 16 ? 100.0 * (lane + 1) : 0.0;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[r * 64 + lane] = c[r];
}
int main() {
  double *d, h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad_a = 0, bad_b = 0;
  for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) {
    const int j = l & 15, blk = l >> 4;
    bad_a += h[r * 64 + l] != 100.0 * (4 * r + blk + 1) * (j + 1);        // i = 4 r + blk   (assumed)
    bad_b += h[r * 64 + l] != 100.0 * (4 * blk + r + 1) * (j + 1);        // i = 4 blk + r   (the f32 16x16x4 layout)
  }
  printf("mfma_f64_16x16x4 D layout: i = 4 * component + lane / 16: %s;  i = 4 * (lane / 16) + component: %s\n", bad_a ? "NO" : "yes", bad_b ? "NO" : "yes");
  return bad_a != 0;
}
