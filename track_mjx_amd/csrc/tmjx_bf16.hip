// csrc/tmjx_bf16.hip — second translation unit of libtmjx_hip.so: the bf16-operand GEMM family (csrc/gemm_bf16.h) and its C-ABI entry points
// (include/tmjx.h "bf16 GEMM-input mode").  Compiled next to tmjx_hip.hip (track_mjx_amd/hip.py:build) and linked into the same library.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <string>

#include "../../include/tmjx.h"
#include "gemm_bf16.h"

extern "C" int tmjx_internal_fail(int code, const char *msg);       // tmjx_hip.hip: records the calling thread's error message
static int fail(int code, const std::string &msg) { return tmjx_internal_fail(code, msg.c_str()); }
#define BZ_ALIGN (4 * sizeof(bz_t) - 1)      // a saved pre-activation row is accessed four elements at a time (bz_t: gemm_bf16.h)
extern "C" int tmjx_bf16_z_bytes(void) { return (int)sizeof(bz_t); }
static int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TMJX_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return TMJX_OK;
}

template <int MI, int NI, int EPI, bool AF32, bool DMA_A>
static int launch_bgemm2(const void *A, int lda, const bf16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K, const BgEpi &epi, hipStream_t s) {
  using Cfg = BgCfg<MI, NI>;
  static bool attr_set = false;            // > 64 KiB of dynamic LDS needs the attribute once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k_bgemm_nt<MI, NI, EPI, AF32, DMA_A>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
    if (e != hipSuccess) return fail(TMJX_EHIP, std::string("hipFuncSetAttribute(k_bgemm_nt): ") + hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid((M + Cfg::BM - 1) / Cfg::BM, (N + Cfg::BN - 1) / Cfg::BN);
  hipLaunchKernelGGL((k_bgemm_nt<MI, NI, EPI, AF32, DMA_A>), grid, dim3(512), Cfg::LDS, s, A, lda, B, ldb, bias, C, ldc, M, N, K, epi);
  return check_launch("k_bgemm_nt");
}
template <int MI, int NI, int EPI, bool AF32>
static int launch_bgemm(const void *A, int lda, const bf16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K, const BgEpi &epi, hipStream_t s) {
  // bf16 activations with a contraction length that is a whole number of K tiles: A by LDS-DMA, two tiles ahead (TMJX_BG_NO_DMA_A=1: A/B switch)
  static const bool no_dma = getenv("TMJX_BG_NO_DMA_A") != nullptr;
  if constexpr (!AF32) {
    if (!(K % BG_BK) && !no_dma) return launch_bgemm2<MI, NI, EPI, false, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  }
  return launch_bgemm2<MI, NI, EPI, AF32, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}
template <int EPI, bool AF32>
static int bgemm_by_width(const void *A, int lda, const bf16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K, const BgEpi &epi, hipStream_t s) {
  if (N <= 128) return launch_bgemm<5, 1, EPI, AF32>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  if (N <= 256) return launch_bgemm<5, 2, EPI, AF32>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  return launch_bgemm<5, 4, EPI, AF32>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}

// rows of M per slab and number of slabs: about two workgroups (four waves each) per CU
static void bdw_split(int M, int N, int K, int *rows_per_split, int *S, int *ld_slab, int max_slabs = 0) {
  const int tiles = ((N + BGDW_BT - 1) / BGDW_BT) * ((K + BGDW_BT - 1) / BGDW_BT);
  static const int target = getenv("TMJX_BDW_WGS") ? atoi(getenv("TMJX_BDW_WGS")) : 512;      // tuning knob
  int want = (target + tiles - 1) / tiles;
  if (max_slabs > 0 && want > max_slabs) want = max_slabs;      // (a problem of a group: the group fills the chip, not the problem)
  if (want < 1) want = 1;
  int rps = (((M + want - 1) / want) + BGDW_BM - 1) / BGDW_BM * BGDW_BM;
  if (rps < BGDW_BM) rps = BGDW_BM;
  *rows_per_split = rps;
  *S = (M + rps - 1) / rps;
  *ld_slab = ((K + BGDW_BT - 1) / BGDW_BT) * BGDW_BT + 4;
}

// ---- fused block epilogues (whole-row tiles: the layer is exactly one tile wide)
static bool row_tile_width(int N) { return N == 128 || N == 256 || N == 512; }
template <int EPI>
static int bgemm_epi(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K, const BgEpi &epi, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a_is_f32) return bgemm_by_width<EPI, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  return bgemm_by_width<EPI, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}
static int check_operands(const char *what, const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, int M, int N, int K) {
  if (!A || !B) return fail(TMJX_EINVAL, std::string(what) + ": null argument");
  if (M < 1 || N < 1 || K < 1 || lda < K || ldb < ((K + 63) & ~63)) return fail(TMJX_EINVAL, std::string(what) + ": bad sizes / leading dimensions (ldb must reach ceil64(K))");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (ldb & 7) || (lda & (a_is_f32 ? 3 : 7))) return fail(TMJX_EINVAL, std::string(what) + ": operand rows must be 16-byte aligned");
  return TMJX_OK;
}

extern "C" {

int tmjx_bf16_shadow(const tmjx_bf16_shadow_t *items, int n, void *stream) {
  if (!items) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || n > BG_SHADOW_MAX) return fail(TMJX_EINVAL, "1 .. 24 matrices per call");
  BgShadowTable T;
  T.n = n;
  int blk = 0;
  for (int i = 0; i < n; i++) {
    const tmjx_bf16_shadow_t &q = items[i];
    if (!q.src || (!q.dst && !q.dst_t)) return fail(TMJX_EINVAL, "null pointer in an item");
    const int Np = (q.N + 63) & ~63, Kp = (q.K + 63) & ~63;
    if (q.N < 1 || q.K < 1 || q.ld_src < q.K || (q.dst && q.ld_dst < Kp) || (q.dst_t && q.ld_dst_t < Np))
      return fail(TMJX_EINVAL, "bad sizes: the shadows' leading dimensions must reach the next multiple of 64");
    T.it[i] = BgShadowItem{q.src, q.dst, q.dst_t, q.N, q.K, q.ld_src, q.ld_dst, q.ld_dst_t, blk};
    blk += (Np / 32) * (Kp / 32);
  }
  hipLaunchKernelGGL(k_bf16_shadow, dim3(blk), dim3(256), 0, (hipStream_t)stream, T);
  return check_launch("k_bf16_shadow");
}

int tmjx_bgemm_nt(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, float *C, int ldc, int M, int N, int K, void *stream) {
  if (!A || !B || !C) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || lda < K || ldc < N || ldb < ((K + 63) & ~63)) return fail(TMJX_EINVAL, "bad sizes / leading dimensions (ldb must reach ceil64(K))");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (ldb & 7) || (lda & (a_is_f32 ? 3 : 7))) return fail(TMJX_EINVAL, "tmjx_bgemm_nt: operand rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  BgEpi epi{};
  static float *stamps = nullptr;          // TMJX_BG_STAMPS=<device pointer, hex>: per-workgroup s_memtime stamps (tools/bf16_stamps.py)
  static bool checked = false;
  if (!checked) { const char *e = getenv("TMJX_BG_STAMPS"); if (e) stamps = (float *)(uintptr_t)strtoull(e, nullptr, 16); checked = true; }
  epi.partial = stamps;
  if (a_is_f32) return bgemm_by_width<0, true>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  return bgemm_by_width<0, false>(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}

int tmjx_bgemm_row_tile_ok(int N) { return row_tile_width(N) ? 1 : 0; }
long long tmjx_bgemm_partial_floats(int M, int N, int sums) { return (M < 1 || N < 1) ? 0 : (long long)((M + 79) / 80) * sums * N; }

int tmjx_bgemm_ln_fwd(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, const float *gamma, const float *beta, uint16_t *Z16, int ldz,
                      uint16_t *Y16, int ldy16, float *stats, int M, int N, int K, float eps, void *stream) {
  if (int rc = check_operands("tmjx_bgemm_ln_fwd", A, a_is_f32, lda, B, ldb, M, N, K)) return rc;
  if (!bias || !gamma || !beta || !Z16 || !Y16 || !stats) return fail(TMJX_EINVAL, "tmjx_bgemm_ln_fwd: null argument");
  if (!row_tile_width(N) || ldz < N || ldy16 < N || (ldz & 3) || (ldy16 & 3) || ((uintptr_t)Z16 & BZ_ALIGN) || ((uintptr_t)Y16 & 7))
    return fail(TMJX_EINVAL, "tmjx_bgemm_ln_fwd: N must be 128, 256 or 512 and the outputs' rows aligned");
  BgEpi e{};
  e.gamma = gamma; e.beta = beta; e.stats = stats; e.y16 = Y16; e.ldy16 = ldy16; e.eps = eps;
  return bgemm_epi<1>(A, a_is_f32, lda, B, ldb, bias, reinterpret_cast<float *>(Z16), ldz, M, N, K, e, stream);      // (EPI 1 / 3: C is the bf16 array)
}

int tmjx_bgemm_ln_bwd(const void *dY, int dy_is_f32, int ldy, const uint16_t *Bt, int ldb, const uint16_t *z, int ldz, const float *bias, const float *gamma,
                      const float *stats, uint16_t *dZ16, int lddz, float *partial, int M, int N, int K, void *stream) {
  if (int rc = check_operands("tmjx_bgemm_ln_bwd", dY, dy_is_f32, ldy, Bt, ldb, M, N, K)) return rc;
  if (!z || !bias || !gamma || !stats || !dZ16 || !partial) return fail(TMJX_EINVAL, "tmjx_bgemm_ln_bwd: null argument");
  if (!row_tile_width(N) || ldz < N || lddz < N || (ldz & 3) || (lddz & 3) || ((uintptr_t)z & BZ_ALIGN) || ((uintptr_t)dZ16 & 7))
    return fail(TMJX_EINVAL, "tmjx_bgemm_ln_bwd: N must be 128, 256 or 512 and z / dZ rows aligned");
  BgEpi e{};
  e.gamma = gamma; e.stats = const_cast<float *>(stats); e.y16 = dZ16; e.ldy16 = lddz; e.z = reinterpret_cast<const bz_t *>(z); e.ldz = ldz; e.partial = partial;
  return bgemm_epi<2>(dY, dy_is_f32, ldy, Bt, ldb, bias, nullptr, N, M, N, K, e, stream);
}

int tmjx_bgemm_silu_fwd(const void *A, int a_is_f32, int lda, const uint16_t *B, int ldb, const float *bias, uint16_t *Z16, int ldz, uint16_t *Y16, int ldy16,
                        float *Yf, int ldyf, int M, int N, int K, void *stream) {
  if (int rc = check_operands("tmjx_bgemm_silu_fwd", A, a_is_f32, lda, B, ldb, M, N, K)) return rc;
  if (!bias || !Z16 || (!Y16 && !Yf)) return fail(TMJX_EINVAL, "tmjx_bgemm_silu_fwd: null argument");
  if (ldz < N || (Y16 && (ldy16 < N || (ldy16 & 3) || ((uintptr_t)Y16 & 7))) || (Yf && ldyf < N)) return fail(TMJX_EINVAL, "tmjx_bgemm_silu_fwd: bad output leading dimensions / alignment");
  BgEpi e{};
  e.y16 = Y16; e.ldy16 = ldy16; e.yf = Yf; e.ldyf = ldyf;
  return bgemm_epi<3>(A, a_is_f32, lda, B, ldb, bias, reinterpret_cast<float *>(Z16), ldz, M, N, K, e, stream);
}

int tmjx_bgemm_silu_bwd(const void *dY, int dy_is_f32, int ldy, const uint16_t *Bt, int ldb, const uint16_t *z, int ldz, const float *bias, uint16_t *dZ16, int lddz,
                        float *partial, int M, int N, int K, void *stream) {
  if (int rc = check_operands("tmjx_bgemm_silu_bwd", dY, dy_is_f32, ldy, Bt, ldb, M, N, K)) return rc;
  if (!z || !bias || !dZ16 || !partial) return fail(TMJX_EINVAL, "tmjx_bgemm_silu_bwd: null argument");
  if (ldz < N || lddz < N || (lddz & 3) || ((uintptr_t)dZ16 & 7) || ((uintptr_t)z & BZ_ALIGN)) return fail(TMJX_EINVAL, "tmjx_bgemm_silu_bwd: bad leading dimensions / alignment");
  BgEpi e{};
  e.y16 = dZ16; e.ldy16 = lddz; e.z = reinterpret_cast<const bz_t *>(z); e.ldz = ldz; e.partial = partial;
  return bgemm_epi<4>(dY, dy_is_f32, ldy, Bt, ldb, bias, nullptr, N, M, N, K, e, stream);
}

int tmjx_bf_silu_bwd(const float *dY, int ldy, const uint16_t *z, int ldz, const float *bias, uint16_t *dZ16, int lddz, float *partial, int M, int N, void *stream) {
  if (!dY || !z || !bias || !dZ16 || !partial) return fail(TMJX_EINVAL, "tmjx_bf_silu_bwd: null argument");
  if (M < 1 || N < 1 || ldy < N || ldz < N || lddz < N) return fail(TMJX_EINVAL, "tmjx_bf_silu_bwd: bad sizes / leading dimensions");
  const bool v4 = !(N & 3) && N <= 1024 && !(ldy & 3) && !(ldz & 3) && !(lddz & 3) && !(((uintptr_t)dY | (uintptr_t)bias) & 15) && !((uintptr_t)dZ16 & 7) && !((uintptr_t)z & BZ_ALIGN);
  if (v4) hipLaunchKernelGGL(k_bf_silu_bwd4<false>, dim3((M + 79) / 80), dim3(256), 0, (hipStream_t)stream, dY, ldy, (const float *)nullptr, (const float *)nullptr, reinterpret_cast<const bz_t *>(z), ldz, bias,
                             dZ16, lddz, partial, M, N);
  else hipLaunchKernelGGL(k_bf_silu_bwd, dim3((M + 79) / 80), dim3(256), 0, (hipStream_t)stream, dY, ldy, reinterpret_cast<const bz_t *>(z), ldz, bias, dZ16, lddz, partial, M, N);
  return check_launch("k_bf_silu_bwd");
}

int tmjx_bf_silu_bwd_rank1(const float *dy1, const float *w1, const uint16_t *z, int ldz, const float *bias, uint16_t *dZ16, int lddz, float *partial, int M, int N, void *stream) {
  if (!dy1 || !w1 || !z || !bias || !dZ16 || !partial) return fail(TMJX_EINVAL, "tmjx_bf_silu_bwd_rank1: null argument");
  if (M < 1 || N < 4 || (N & 3) || N > 1024 || ldz < N || lddz < N || (ldz & 3) || (lddz & 3) || (((uintptr_t)w1 | (uintptr_t)bias) & 15) || ((uintptr_t)dZ16 & 7) || ((uintptr_t)z & BZ_ALIGN))
    return fail(TMJX_EINVAL, "tmjx_bf_silu_bwd_rank1: N must be a multiple of 4 up to 1024, rows 16-byte aligned");
  hipLaunchKernelGGL(k_bf_silu_bwd4<true>, dim3((M + 79) / 80), dim3(256), 0, (hipStream_t)stream, (const float *)nullptr, 0, dy1, w1, reinterpret_cast<const bz_t *>(z), ldz, bias, dZ16, lddz, partial, M, N);
  return check_launch("k_bf_silu_bwd_rank1");
}

long long tmjx_bgemm_dw_scratch_floats(int M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  int rps, S, ld;
  bdw_split(M, N, K, &rps, &S, &ld);
  return (long long)S * N * ld;
}

int tmjx_bgemm_dw(const void *dY, int y_is_f32, int ldy, const void *X, int x_is_f32, int ldx, float *dW, int lddw, float *db, float *scratch,
                  int M, int N, int K, void *stream) {
  if (!dY || !X || !dW || !scratch) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1 || ldy < N || ldx < K || lddw < K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions");
  if (((uintptr_t)dY & 15) || ((uintptr_t)X & 15) || (ldy & (y_is_f32 ? 3 : 7)) || (ldx & (x_is_f32 ? 3 : 7)))
    return fail(TMJX_EINVAL, "tmjx_bgemm_dw: operand rows must be 16-byte aligned");
  int rps, S, ld;
  bdw_split(M, N, K, &rps, &S, &ld);
  hipStream_t s = (hipStream_t)stream;
  const int tn = (N + BGDW_BT - 1) / BGDW_BT, tk = (K + BGDW_BT - 1) / BGDW_BT;
  dim3 grid((unsigned)(tn * tk * ((S + 7) / 8) * 8));
  const size_t lds = 2 * BGDW_STAGE;
  const int wb = db ? 1 : 0;
  if (y_is_f32 && x_is_f32) hipLaunchKernelGGL((k_bgemm_dw<true, true>), grid, dim3(256), lds, s, dY, ldy, X, ldx, scratch, M, N, K, wb, rps, ld, tn, tk, S);
  else if (y_is_f32) hipLaunchKernelGGL((k_bgemm_dw<true, false>), grid, dim3(256), lds, s, dY, ldy, X, ldx, scratch, M, N, K, wb, rps, ld, tn, tk, S);
  else if (x_is_f32) hipLaunchKernelGGL((k_bgemm_dw<false, true>), grid, dim3(256), lds, s, dY, ldy, X, ldx, scratch, M, N, K, wb, rps, ld, tn, tk, S);
  else hipLaunchKernelGGL((k_bgemm_dw<false, false>), grid, dim3(256), lds, s, dY, ldy, X, ldx, scratch, M, N, K, wb, rps, ld, tn, tk, S);
  const long long total = (long long)N * (K + wb);
  hipLaunchKernelGGL(k_bgemm_dw_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float *)scratch, dW, db, S, N, K, wb, ld, lddw);
  return check_launch("k_bgemm_dw");
}

// Every weight gradient of a backward pass in one launch + one reduction launch (k_bgemm_dw_grouped): problems as tmjx_bgemm_dw takes them, each with a
// scratch of tmjx_bgemm_dw_scratch_floats(M, N, K) floats.  target_wgs: workgroups the GROUP should bring (0: TMJX_BDW_GROUP_WGS, default 2048).
int tmjx_bgemm_dw_grouped(const tmjx_bdw_problem_t *probs, int n, int target_wgs, void *stream) {
  if (!probs) return fail(TMJX_EINVAL, "null argument");
  if (n < 1 || n > BDW_GROUP_MAX) return fail(TMJX_EINVAL, "1 .. 24 problems per group");
  static const int group_default = getenv("TMJX_BDW_GROUP_WGS") ? atoi(getenv("TMJX_BDW_GROUP_WGS")) : 2048;
  const int group_target = target_wgs > 0 ? target_wgs : group_default;
  int all_tiles = 0;
  for (int i = 0; i < n; i++) all_tiles += ((probs[i].N + BGDW_BT - 1) / BGDW_BT) * ((probs[i].K + BGDW_BT - 1) / BGDW_BT);
  // slabs in EIGHTS: a problem's slab s runs on XCD s % 8 (k_bgemm_dw's order: the tiles of one row slab share an L2), so 6 slabs leave two XCDs without
  // work (measured on config 5's 17 layers: 6 slabs 4.05 ms per minibatch step, 7 -> 3.98 with a 16-slab grid, 8 -> 3.74; layer by layer 3.80)
  int max_slabs = ((group_target + all_tiles / 2) / (all_tiles > 0 ? all_tiles : 1) + 4) / 8 * 8;
  if (max_slabs < 8) max_slabs = 8;
  BdwGroup G;
  G.n = n;
  int wg = 0, red = 0;
  for (int i = 0; i < n; i++) {
    const tmjx_bdw_problem_t &q = probs[i];
    if (!q.dY || !q.X || !q.dW || !q.scratch) return fail(TMJX_EINVAL, "null pointer in a problem");
    if (q.M < 1 || q.N < 1 || q.K < 1 || q.ldy < q.N || q.ldx < q.K || q.lddw < q.K) return fail(TMJX_EINVAL, "bad sizes / leading dimensions in a problem");
    if (((uintptr_t)q.dY & 15) || ((uintptr_t)q.X & 15) || (q.ldy & (q.y_is_f32 ? 3 : 7)) || (q.ldx & (q.x_is_f32 ? 3 : 7)))
      return fail(TMJX_EINVAL, "tmjx_bgemm_dw_grouped: operand rows must be 16-byte aligned");
    BdwProblem &P = G.p[i];
    P.dY = q.dY; P.X = q.X; P.dW = q.dW; P.db = q.db; P.slabs = q.scratch;
    P.ldy = q.ldy; P.ldx = q.ldx; P.lddw = q.lddw; P.M = q.M; P.N = q.N; P.K = q.K; P.y_f32 = q.y_is_f32 ? 1 : 0; P.x_f32 = q.x_is_f32 ? 1 : 0;
    bdw_split(q.M, q.N, q.K, &P.rows_per_split, &P.S, &P.ld_slab, max_slabs);
    P.tiles_n = (q.N + BGDW_BT - 1) / BGDW_BT; P.tiles_k = (q.K + BGDW_BT - 1) / BGDW_BT;
    P.wg_begin = wg; wg += P.tiles_n * P.tiles_k * ((P.S + 7) / 8) * 8;
    P.red_begin = red; red += (int)(((long long)q.N * (q.K + (q.db ? 1 : 0)) + 255) / 256);
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bgemm_dw_grouped, dim3(wg), dim3(256), 2 * BGDW_STAGE, s, G);
  hipLaunchKernelGGL(k_bgemm_dw_reduce_grouped, dim3(red), dim3(256), 0, s, G);
  return check_launch("k_bgemm_dw_grouped");
}

}  // extern "C"

int tmjx_linear_nolds_bf16(const float *A, int64_t sa_row, int64_t sa_k, const uint16_t *W, int ldw, const float *bias, float *C, int M, int N, int K,
                           const float *mean, const float *inv_std, void *stream) {
  if (!A || !W || !C) return fail(TMJX_EINVAL, "null argument");
  if (M < 1 || N < 1 || K < 1) return fail(TMJX_EINVAL, "bad shape");
  const bool kmajor = sa_k != 1;
  if ((K & 3) || (ldw & 63) || ldw < K || ((uintptr_t)W & 15) || (!kmajor && ((sa_row & 3) || ((uintptr_t)A & 15))) || (mean && (((uintptr_t)mean | (uintptr_t)inv_std) & 15)) || (mean && !inv_std))
    return fail(TMJX_EINVAL, "tmjx_linear_nolds_bf16: K % 4 == 0, a bf16 shadow with rows padded to a multiple of 64 (tmjx_bf16_shadow) and 16-byte aligned operands");
  dim3 grid((M + 31) / 32, (N + 31) / 32);
  hipStream_t s = (hipStream_t)stream;
  if (kmajor) hipLaunchKernelGGL((k_linear_nolds_bf16<true>), grid, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, (const bf16_t *)W, ldw, bias, C, M, N, K, mean, inv_std);
  else hipLaunchKernelGGL((k_linear_nolds_bf16<false>), grid, dim3(64), 0, s, A, (long long)sa_row, (long long)sa_k, (const bf16_t *)W, ldw, bias, C, M, N, K, mean, inv_std);
  return check_launch("k_linear_nolds_bf16");
}
