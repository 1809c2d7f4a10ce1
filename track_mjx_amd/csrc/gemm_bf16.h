// csrc/gemm_bf16.h — the learner's dense contractions with bf16 operands / fp32 accumulation on v_mfma_f32_16x16x32_bf16 (BASELINE config 5:
// "bf16 MLP on MFMA"; reference layers: flax nn.Dense in track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76, sizes
// track_mjx/config/rodent-full-clips.yaml:50-57, brax value MLP ppo_networks.py:180-184).  Same shapes as csrc/gemm_kernels.h — a tall activation
// matrix (40 960 rows in config 5) against small weight matrices — but the matrix pipe is 16 x faster here, so the kernels are built around
// the memory system instead:
//
//   * NO cast pass.  Activations arrive as fp32 (converted to bf16 when the staged registers are written to LDS: v_cvt_pk_bf16_f32, round to
//     nearest even = torch's .to(bfloat16)) or as bf16 written by the producing kernel's epilogue; weights come from bf16 SHADOWS of the fp32
//     master parameters (k_bf16_shadow: [N][ceil64 K] and the transpose [K][ceil64 N], zero padded, refreshed once per optimiser step) and go
//     global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write).
//   * k_bgemm_nt: C[M][N] = A[M][K] . B[N][K]^T for both the forward pass (B = W shadow) and the input gradient (B = W^T shadow): every
//     fragment is one ds_read_b128 from a [row][64] image whose 16-byte chunks are XOR-swizzled (chunk ^= (row >> 1) & 7: the 16 lanes of a
//     ds_read_b128 group hit 16 distinct slots of the 256-byte bank row).  Workgroup tile 16 MI rows x 128 NI columns, 8 waves side by side
//     along N; the weight fragment is the MFMA's FIRST operand, so a lane ends up with four consecutive columns of one row (vector stores).
//   * k_bgemm_dw: dW[N][K] = dY[M][N]^T . X[M][K], the contraction runs over the rows.  Both operands are staged ROW-major as they lie in
//     memory and read TRANSPOSED by ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major): no transposed copy
//     of any activation exists anywhere.  Slabs over row ranges + csrc/gemm_kernels.h's k_dw_reduce, the bias gradient rides along.
#pragma once
#include "silu_math.h"

typedef short bgs8 __attribute__((ext_vector_type(8)));
typedef short bgs4 __attribute__((ext_vector_type(4)));
typedef __bf16 bgb8 __attribute__((ext_vector_type(8)));
typedef __bf16 bgb2 __attribute__((ext_vector_type(2)));
typedef float bgf4 __attribute__((ext_vector_type(4)));
typedef float bgf2 __attribute__((ext_vector_type(2)));
typedef unsigned bgu4 __attribute__((ext_vector_type(4)));
typedef unsigned bgu2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;
// (the acting policy's layer raises its waves' priority over the physics waves it runs next to: csrc/tm_common.h, TM_PRIO_ACTING)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TMJX_NO_ACT_PRIO)
#define TM_PRIO_ACTING_BF() __builtin_amdgcn_s_setprio(3)
#else
#define TM_PRIO_ACTING_BF() do { } while (0)
#endif

#define BG_LDS(p) ((__attribute__((address_space(3))) void *)(p))
#define BG_GLB(p) ((const __attribute__((address_space(1))) void *)(p))
#define BG_BK 64

// two floats -> two bf16 in one dword (lo in the low half), round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned bg_pack(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(bgf2{lo, hi}, bgb2)); }
__device__ __forceinline__ bgu4 bg_pack8(bgf4 a, bgf4 b) { return bgu4{bg_pack(a.x, a.y), bg_pack(a.z, a.w), bg_pack(b.x, b.y), bg_pack(b.z, b.w)}; }
__device__ __forceinline__ float bg_f32(bf16_t h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// four bf16 (two dwords, element 0 in the low half of the first) -> four floats
__device__ __forceinline__ bgf4 bg_unpack4(bgu2 u) {
  return bgf4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u), __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
}
__device__ __forceinline__ bgu2 bg_pack4(bgf4 v) { return bgu2{bg_pack(v.x, v.y), bg_pack(v.z, v.w)}; }
// The saved pre-activations z of a block (what the backward epilogues re-read): bf16 — BASELINE config 5 is a "bf16 MLP on MFMA" — or, with
// -DTMJX_BF16_Z_F32, fp32 (SURVEY a16's narrower reading: "bf16 only for GEMM inputs"; round 3's form of these kernels).  One element type and
// four accessors; the C-ABI's `Z16` / `z16` pointers then point to floats (tmjx_bf16_z_bytes() says which: 2 or 4).
#ifdef TMJX_BF16_Z_F32
typedef float bz_t;
__device__ __forceinline__ bgf4 bz_load4(const bz_t *p) { return *reinterpret_cast<const bgf4 *>(p); }
__device__ __forceinline__ void bz_store4(bz_t *p, bgf4 v) { *reinterpret_cast<bgf4 *>(p) = v; }
__device__ __forceinline__ float bz_f32(bz_t h) { return h; }
__device__ __forceinline__ bz_t bz_from(float v) { return v; }
#else
typedef bf16_t bz_t;
__device__ __forceinline__ bgf4 bz_load4(const bz_t *p) { return bg_unpack4(*reinterpret_cast<const bgu2 *>(p)); }
__device__ __forceinline__ void bz_store4(bz_t *p, bgf4 v) { *reinterpret_cast<bgu2 *>(p) = bg_pack4(v); }
__device__ __forceinline__ float bz_f32(bz_t h) { return bg_f32(h); }
__device__ __forceinline__ bz_t bz_from(float v) { return (bf16_t)(bg_pack(v, 0.f) & 0xffffu); }
#endif

// byte offset of 16-byte chunk c (0 .. 7) of row r in a [rows][64 bf16] LDS image (128-byte rows, chunks XOR-swizzled)
__device__ __forceinline__ int bg_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// ---- bf16 shadows of the weights: dst[n][k] (leading dimension ldd >= ceil64(K), zero beyond K) and dstT[k][n] (ldt >= ceil64(N), zero beyond N)
#define BG_SHADOW_MAX 24
struct BgShadowItem { const float *src; bf16_t *dst, *dstT; int N, K, lds_, ldd, ldt, blk_begin; };
struct BgShadowTable { BgShadowItem it[BG_SHADOW_MAX]; int n; };
// one workgroup (256 threads) per 32 x 32 tile of the padded [ceil64 N][ceil64 K] index space of one matrix
__global__ __launch_bounds__(256) void k_bf16_shadow(const BgShadowTable T) {
  __shared__ float tile[32][33];
  int i = 0;
  for (int j = 1; j < T.n; j++) if ((int)blockIdx.x >= T.it[j].blk_begin) i = j;
  const BgShadowItem &q = T.it[i];
  const int Np = (q.N + 63) & ~63, Kp = (q.K + 63) & ~63, tk = Kp / 32;
  const int local = blockIdx.x - q.blk_begin, n0 = (local / tk) * 32, k0 = (local % tk) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    const float v = (n < q.N && k < q.K) ? q.src[(size_t)n * q.lds_ + k] : 0.f;
    tile[r][tx] = v;
    if (q.dst && n < q.N) q.dst[(size_t)n * q.ldd + k] = (bf16_t)(bg_pack(v, 0.f) & 0xffffu);
  }
  __syncthreads();
  if (q.dstT) {
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int k = k0 + r, n = n0 + tx;
      if (k < q.K) q.dstT[(size_t)k * q.ldt + n] = (bf16_t)(bg_pack(tile[tx][r], 0.f) & 0xffffu);
    }
  }
}

// ---- C = A . B^T ---------------------------------------------------------------------------------------------------------------------
// Epilogues.  Register r of accumulator tile (a, b) of lane (li = lane & 15, kq = lane >> 4) of wave w is C[m0 + 16 a + li][n0 + 128 NI... see below]
struct BgEpi {
  const float *gamma, *beta;      // EPI 1 / 2: LayerNorm scale / shift of the block
  float *stats;                   // EPI 1 writes (mean, 1 / std) per row, EPI 2 reads them
  bf16_t *y16; int ldy16;         // EPI 1 / 3: the activation as bf16 (the next layer's operand);  EPI 2 / 4: d loss / d z as bf16
  float *yf; int ldyf;            // EPI 3: the activation as fp32 instead (the consumer is an fp32 kernel)
  const bz_t *z; int ldz;         // EPI 2 / 4: the block's saved pre-activation (as EPI 1 / 3 stored it — bz_t: bf16 or, -DTMJX_BF16_Z_F32, fp32 — without the bias)
  float *partial;                 // EPI 2 / 4: per-row-tile column sums: [gridDim.x][3][N] (d gamma | d beta | d bias) resp. [gridDim.x][N] (d bias)
  float eps;
};
// sum over the 16 lanes of a DPP row (all 16 lanes get it): quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float bg_row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
  return x;
}

template <int MI, int NI> struct BgCfg {
  // LDS: two stages of the activation tile in front of two stages of the weight tile: 2 x 10 KiB + 2 x 64 KiB = 148 KiB at the widest tile
  static constexpr int BM = 16 * MI, BN = 128 * NI, A_BYTES = BM * 128, B_BYTES = BN * 128, B_BASE = 2 * A_BYTES, LDS = 2 * A_BYTES + 2 * B_BYTES;
};

// EPI 0: C = acc (+ bias), fp32.
// EPI 1: Dense -> SiLU -> LayerNorm forward (the tile spans whole rows: N == BN): z = acc to C AS BF16 (C is a bf16 array here, ldc in elements;
//        without the bias: what the backward kernels expect), y = LayerNorm(silu(z + bias)) — from the fp32 accumulators — as bf16 to y16,
//        (mean, 1 / std) to stats.  The pre-activation is the block's only fp32-sized store: as bf16 ("bf16 MLP": BASELINE config 5) the forward
//        kernels write 4 instead of 6 bytes per element and the backward epilogues read 2 instead of 4, twice.
// EPI 2: the tile is d loss / d y of a Dense -> SiLU -> LayerNorm block (whole rows, N == BN = the block's width; this GEMM is the input
//        gradient of the block's consumer): that block's LayerNorm + SiLU backward (the arithmetic of k_silu_ln_bwd) applied on the
//        accumulators, d loss / d z stored as bf16 (the operand of the block's own input- and weight-gradient GEMMs) + the row tile's column
//        sums (dy ahat | dy | dz).  C is not written.
// EPI 3: Dense -> SiLU forward (brax value MLP): z = acc to C as bf16 (as EPI 1), y = silu(z + bias) as bf16 (or fp32).   EPI 4: its backward on the tile
//        d loss / d y: dz = dy silu'(z + bias) as bf16 + column sums of dz.
// DMA_A (bf16 A, K a multiple of 64): the activation tile goes global -> LDS by LDS-DMA as well; otherwise it is staged through registers (fp32 A:
// converted at the LDS write; ragged K: masked).
// K loop: ONE barrier per 64-deep tile, in its MIDDLE, with 20 MFMAs per wave on either side of it (the structure of csrc/gemm_kernels.h):
//   first half   fragments of the tile's second k step are read from LDS while the first step's MFMAs run (their fragments were read a half
//                tile earlier); register-staged A of the NEXT tile is written to the other stage;
//   barrier      the next tile (DMAs issued a whole tile earlier) has landed, everyone is done reading this tile's stage;
//   second half  the DMAs of the tile after next go into the stage just released, the next tile's first-step fragments are read while the
//                second step's MFMAs run.
// No MFMA ever waits for an LDS read issued right in front of it (the first version read 9 fragments, waited, ran 20 MFMAs, twice per tile: the
// matrix pipe was busy half of the K loop's cycles, tools/bf16_stamps.py).
template <int MI, int NI, int EPI, bool AF32, bool DMA_A = false>
__global__ __launch_bounds__(512) void k_bgemm_nt(const void *__restrict__ Av, int lda, const bf16_t *__restrict__ B, int ldb, const float *__restrict__ bias,
                                                  float *__restrict__ C, int ldc, int M, int N, int K, BgEpi epi) {
  using Cfg = BgCfg<MI, NI>;
  constexpr int BM = Cfg::BM, BN = Cfg::BN, A_BYTES = Cfg::A_BYTES, B_BYTES = Cfg::B_BYTES, B_BASE = Cfg::B_BASE;
  static_assert(!(DMA_A && AF32), "LDS-DMA cannot convert");
  extern __shared__ __attribute__((aligned(16))) char bg_lds[];
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, nw = wave * 16 * NI;
  const int nk = (K + BG_BK - 1) / BG_BK;
  // tuning aid (EPI 0 with a `partial` buffer: tools/bf16_stamps.py): s_memtime stamps of wave 0 around prologue / K loop / stores
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  const bool stamp = EPI == 0 && epi.partial != nullptr;
  if (stamp) ts0 = __builtin_amdgcn_s_memtime();

  // ---- B: LDS-DMA.  One wave instruction moves 1 KiB = 8 rows of the image; wave w moves row groups w, w + 8, ...  The LDS side is
  // lane-linear (lane i -> bytes 16 i .. 16 i + 15 of the group: row i >> 3, PHYSICAL chunk i & 7), so the swizzle goes on the source
  // address: the lane fetches the LOGICAL chunk (i & 7) ^ swizzle(row).  Rows beyond N are clamped (their columns are never stored).
  constexpr int B_INSTR = BN / 64;
  const bf16_t *bsrc[B_INSTR];
#pragma unroll
  for (int i = 0; i < B_INSTR; i++) {
    const int row = 8 * (wave + 8 * i) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    bsrc[i] = B + (size_t)min(n0 + row, N - 1) * ldb + 8 * c;
  }
  auto issue_b = [&](int stage, int k0) {
#pragma unroll
    for (int i = 0; i < B_INSTR; i++)
      __builtin_amdgcn_global_load_lds(BG_GLB(bsrc[i] + k0), BG_LDS(bg_lds + B_BASE + stage * B_BYTES + (wave + 8 * i) * 1024), 16, 0, 0);
  };

  // ---- A: registers (fp32 -> bf16 at the LDS write, or bf16 as it is), in units of 16 BYTES OF MEMORY so that a wave instruction reads
  // whole contiguous row segments (a lane reading 32 bytes = two dwordx4 with a 32-byte lane stride half-uses every cache line per
  // instruction: the vector cache's fill rate, ~29 B/clk/CU, is what bounds these kernels).  fp32: unit = 4 elements -> 8 bytes of LDS;
  // bf16: unit = 8 elements -> 16 bytes.  Unit f = t + 512 p: row f / UPR, unit f % UPR of the row.
  constexpr int UPR = AF32 ? 16 : 8, EPU = 64 / UPR;                  // units per 64-element row of the tile, elements per unit
  constexpr int A_UNITS = BM * UPR, A_FULL = A_UNITS / 512, A_REM_WAVES = (A_UNITS % 512) / 64, A_PASS = A_FULL + (A_REM_WAVES ? 1 : 0);
  static_assert(A_UNITS % 64 == 0, "whole waves");
  struct AStage { bgu4 v[A_PASS]; };
  const char *arow[A_PASS];
#pragma unroll
  for (int p = 0; p < A_PASS; p++) {
    const int f = t + 512 * p, r = min(f / UPR, BM - 1), u = f % UPR;
    arow[p] = (const char *)Av + ((size_t)min(m0 + r, M - 1) * lda + EPU * u) * (AF32 ? 4 : 2);
  }
  auto load_a = [&](AStage &R, int k0) {
    const bool fast = k0 + BG_BK <= K;
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      if (p >= A_FULL && wave >= A_REM_WAVES) continue;           // (wave-uniform: the partly filled last pass)
      const int k = k0 + EPU * ((t + 512 * p) % UPR);             // first element of the unit
      // the last, partial K tile: a unit that starts inside the row's allocation (lda) is loaded whole — lda is a multiple of the unit —
      // and its elements beyond K are zeroed below; units beyond lda are not loaded at all
      bgu4 raw = {0u, 0u, 0u, 0u};
      if (fast || k < lda) raw = *reinterpret_cast<const bgu4 *>(arow[p] + (size_t)k0 * (AF32 ? 4 : 2));
      if (!fast) {
        if (AF32) {
#pragma unroll
          for (int j = 0; j < 4; j++) if (k + j >= K) raw[j] = 0u;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (k + 2 * j >= K) raw[j] = 0u;
            else if (k + 2 * j + 1 >= K) raw[j] &= 0xffffu;
          }
        }
      }
      R.v[p] = raw;
    }
  };
  auto write_a = [&](const AStage &R, int stage) {
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      if (p >= A_FULL && wave >= A_REM_WAVES) continue;
      const int f = t + 512 * p, r = f / UPR, u = f % UPR;
      char *dst = bg_lds + stage * A_BYTES;
      if (AF32) {
        const bgf4 v = __builtin_bit_cast(bgf4, R.v[p]);
        *reinterpret_cast<bgu2 *>(dst + bg_off(r, u >> 1) + 8 * (u & 1)) = bgu2{bg_pack(v.x, v.y), bg_pack(v.z, v.w)};
      } else {
        *reinterpret_cast<bgu4 *>(dst + bg_off(r, u)) = R.v[p];
      }
    }
  };

  bgf4 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; a++)
#pragma unroll
    for (int b = 0; b < NI; b++) acc[a][b] = bgf4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses: rows 16 a + li resp. nw + 16 b + li, so the swizzle term depends on li only
  const int sw = (li >> 1) & 7, frow = li * 128;

  struct Frag { bgs8 a[MI], b[NI]; };
  auto fread = [&](Frag &F, int stage, int s) {
    const char *sa = bg_lds + stage * A_BYTES + frow, *sb = bg_lds + B_BASE + stage * B_BYTES + nw * 128 + frow;
    const int x = ((4 * s + kq) ^ sw) << 4;
#pragma unroll
    for (int a = 0; a < MI; a++) F.a[a] = *reinterpret_cast<const bgs8 *>(sa + a * 2048 + x);
#pragma unroll
    for (int b = 0; b < NI; b++) F.b[b] = *reinterpret_cast<const bgs8 *>(sb + b * 2048 + x);
  };
  auto mma = [&](const Frag &F) {
#pragma unroll
    for (int a = 0; a < MI; a++)
#pragma unroll
      for (int b = 0; b < NI; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bgb8, F.b[b]), __builtin_bit_cast(bgb8, F.a[a]), acc[a][b], 0, 0, 0);
  };
  // A by LDS-DMA: BM / 8 wave instructions per tile (8 rows each); wave w issues instruction w, and w + 8 if that exists
  constexpr int A_INSTR = BM / 8, A_EXTRA = A_INSTR > 8 ? A_INSTR - 8 : 0;
  static_assert(!DMA_A || (A_INSTR <= 16 && A_INSTR >= 8), "one or two A instructions per wave and tile");
  const bf16_t *asrc[2] = {nullptr, nullptr};
  if constexpr (DMA_A) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = 8 * (wave + 8 * i) + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      asrc[i] = reinterpret_cast<const bf16_t *>(Av) + (size_t)min(m0 + min(row, BM - 1), M - 1) * lda + 8 * c;
    }
  }
  AStage R;
  // fetch tile `kt` (clamped to the last one: past the end the last tile is fetched once more, into a stage nobody reads any more)
  auto fetch = [&](int stage, int kt) {
    const int k0 = min(kt, nk - 1) * BG_BK;
    issue_b(stage, k0);
    if constexpr (DMA_A) {
      __builtin_amdgcn_global_load_lds(BG_GLB(asrc[0] + k0), BG_LDS(bg_lds + stage * A_BYTES + wave * 1024), 16, 0, 0);
      if (wave < A_EXTRA) __builtin_amdgcn_global_load_lds(BG_GLB(asrc[1] + k0), BG_LDS(bg_lds + stage * A_BYTES + (wave + 8) * 1024), 16, 0, 0);
    } else load_a(R, k0);
  };
  Frag F0, F1;
  fetch(0, 0);
  if constexpr (!DMA_A) write_a(R, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (stamp) ts1 = __builtin_amdgcn_s_memtime();
  fread(F0, 0, 0);
  fetch(1, 1);
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    fread(F1, cur, 1);
    if constexpr (!DMA_A) write_a(R, cur ^ 1);          // tile kt + 1 (loaded a half tile ago) into the other stage: last read before the previous barrier
    mma(F0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt + 1 has landed (LDS-DMA counts on vmcnt)
    __syncthreads();
    fetch(cur, kt + 2);                                 // into the stage everyone has just finished reading
    fread(F0, cur ^ 1, 0);
    mma(F1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (unused) tiles fetched past the end must have landed before the epilogue reuses LDS
  __syncthreads();
  if (stamp) ts2 = __builtin_amdgcn_s_memtime();
  // (an opaque copy of the lane's row / column origin: otherwise the epilogue's MI NI 64-bit store / load addresses are computed in front of the
  // K loop and stay live through it)
  int li_e = li, kq_e = kq;
  asm volatile("" : "+v"(li_e), "+v"(kq_e));
#define li li_e
#define kq kq_e
  // register r of tile (a, b): C[m0 + 16 a + li][n0 + nw + 16 b + 4 kq + r] — four consecutive columns of one row
  if constexpr (EPI == 0) {
    const bool vec = !(ldc & 3) && !((uintptr_t)C & 15);
#pragma unroll
    for (int b = 0; b < NI; b++) {
      const int col = n0 + nw + 16 * b + 4 * kq;
      bgf4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; r++) bv[r] = col + r < N ? bias[col + r] : 0.f;
      }
#pragma unroll
      for (int a = 0; a < MI; a++) {
        const int row = m0 + 16 * a + li;
        if (row < M) {
          float *o = C + (size_t)row * ldc + col;
          const bgf4 v = acc[a][b] + bv;
          if (vec && col + 3 < N) *reinterpret_cast<bgf4 *>(o) = v;
          else {
#pragma unroll
            for (int r = 0; r < 4; r++) if (col + r < N) o[r] = v[r];
          }
        }
      }
    }
    if (stamp && t == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the stores have been accepted by the memory system)
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      float *o = epi.partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
      o[0] = (float)(ts0 & 0xffffffu); o[1] = (float)(ts1 - ts0); o[2] = (float)(ts2 - ts1); o[3] = (float)(ts3 - ts2);
    }
  }
  // cross-wave exchange of per-row partial sums through LDS (the K loop's stages are dead: its last trip ended with a barrier)
  constexpr int NW = 8;
  if constexpr (EPI == 1) {
    // whole rows: N == BN, n0 == 0.  Row sums: over r and b in registers, over kq by two cross-lane adds, over the waves through LDS.
    float *red1 = reinterpret_cast<float *>(bg_lds), *red2 = red1 + BM * NW;
    bgf4 bv[NI], gv[NI], bev[NI];
#pragma unroll
    for (int b = 0; b < NI; b++) {
      const int col = nw + 16 * b + 4 * kq;
      bv[b] = *reinterpret_cast<const bgf4 *>(bias + col); gv[b] = *reinterpret_cast<const bgf4 *>(epi.gamma + col); bev[b] = *reinterpret_cast<const bgf4 *>(epi.beta + col);
    }
    float stat[MI];
#pragma unroll
    for (int a = 0; a < MI; a++) {
      const int row = m0 + 16 * a + li;
      float p = 0.f;
#pragma unroll
      for (int b = 0; b < NI; b++) {
        if (row < M) bz_store4(reinterpret_cast<bz_t *>(C) + (size_t)row * ldc + nw + 16 * b + 4 * kq, acc[a][b]);
#pragma unroll
        for (int r = 0; r < 4; r++) { const float v = acc[a][b][r] + bv[b][r]; acc[a][b][r] = tm_silu(v); p += acc[a][b][r]; }
      }
      p += __shfl_xor(p, 16);
      stat[a] = p + __shfl_xor(p, 32);
    }
    auto exchange = [&](float *red) {  // stat[a] <- sum over the waves
      if (kq == 0) {
#pragma unroll
        for (int a = 0; a < MI; a++) red[(16 * a + li) * NW + wave] = stat[a];
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < MI; a++) {
        const bgf4 *q = reinterpret_cast<const bgf4 *>(red + (16 * a + li) * NW);
        const bgf4 v0 = q[0], v1 = q[1];
        stat[a] = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
      }
    };
    exchange(red1);
    const float inv_n = 1.f / (float)BN;
    float mean[MI];
#pragma unroll
    for (int a = 0; a < MI; a++) {
      mean[a] = stat[a] * inv_n;
      float q = 0.f;
#pragma unroll
      for (int b = 0; b < NI; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) { acc[a][b][r] -= mean[a]; q += acc[a][b][r] * acc[a][b][r]; }      // centred from here on
      q += __shfl_xor(q, 16);
      stat[a] = q + __shfl_xor(q, 32);
    }
    exchange(red2);
#pragma unroll
    for (int a = 0; a < MI; a++) {
      const int row = m0 + 16 * a + li;
      const float rstd = rsqrtf(stat[a] * inv_n + epi.eps);
      if (row < M) {
#pragma unroll
        for (int b = 0; b < NI; b++) {
          const bgf4 y = acc[a][b] * rstd * gv[b] + bev[b];
          *reinterpret_cast<bgu2 *>(epi.y16 + (size_t)row * epi.ldy16 + nw + 16 * b + 4 * kq) = bgu2{bg_pack(y.x, y.y), bg_pack(y.z, y.w)};
        }
        if (wave == 0 && kq == 0) { epi.stats[2 * (size_t)row] = mean[a]; epi.stats[2 * (size_t)row + 1] = rstd; }
      }
    }
  }
  if constexpr (EPI == 2) {
    float *red = reinterpret_cast<float *>(bg_lds);                  // [BM rows][NW][2]
    // (bias / gamma of a column block are re-loaded where they are used — L1 hits — instead of 8 NI registers held through both passes)
    const float inv_n = 1.f / (float)BN;
    float m1[MI], m2[MI];                            // (mean / rstd of a row are re-loaded in the second pass)        // (z is loaded twice, once per pass: keeping it live costs MI NI 4 registers)
#pragma unroll
    for (int a = 0; a < MI; a++) {
      const int row = m0 + 16 * a + li;
      const bool ok = row < M;
      const size_t rr = ok ? row : M - 1;
      const float mean_a = epi.stats[2 * rr], rstd_a = epi.stats[2 * rr + 1];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int b = 0; b < NI; b++) {
        const int col = nw + 16 * b + 4 * kq;
        const bgf4 z4 = bz_load4(epi.z + rr * epi.ldz + col);
        const bgf4 bv4 = *reinterpret_cast<const bgf4 *>(bias + col), gv4 = *reinterpret_cast<const bgf4 *>(epi.gamma + col);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          // (rows past M: clamped duplicates of the last row — nothing of them may reach the row / column sums.  Masked where it is used, in both
          // passes: zeroing the accumulator in place made the compiler keep the masked AND the unmasked copy of some elements across the whole
          // epilogue — 8 spilled registers at NI = 4)
          const float v = z4[r] + bv4[r], dy = ok ? acc[a][b][r] : 0.f;
          const float sig = tm_sigmoid(v), ah = (v * sig - mean_a) * rstd_a, da = dy * gv4[r];
          s1 += da; s2 += da * ah;
        }
      }
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      m1[a] = s1 + __shfl_xor(s1, 32); m2[a] = s2 + __shfl_xor(s2, 32);
      asm volatile("" ::: "memory");            // one row tile's z at a time: hoisting all MI NI loads spills (77 registers at NI = 4)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kq == 0) {
#pragma unroll
      for (int a = 0; a < MI; a++) *reinterpret_cast<bgf2 *>(red + ((16 * a + li) * NW + wave) * 2) = bgf2{m1[a], m2[a]};
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MI; a++) {
      const bgf4 *q = reinterpret_cast<const bgf4 *>(red + (16 * a + li) * NW * 2);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NW / 2; w4++) { const bgf4 x = q[w4]; t1 += x.x + x.z; t2 += x.y + x.w; }
      m1[a] = t1 * inv_n; m2[a] = t2 * inv_n;
    }
    // second pass column block by column block: the three column sums of ONE block are live at a time (all NI of them cost 12 NI registers).
    // z is RE-LOADED through an opaque pointer: otherwise the loads (and the sigmoid / ahat values computed from them) of the first pass are
    // kept live across the exchange for the second one — MI NI 12 registers, spilled at NI = 4
    const bz_t *z2 = epi.z;
    asm volatile("" : "+s"(z2));
    float *pp = epi.partial + (size_t)blockIdx.x * 3 * BN;
#pragma unroll
    for (int b = 0; b < NI; b++) {
      bgf4 cg = {0.f, 0.f, 0.f, 0.f}, cb = cg, cz = cg;
      const int col = nw + 16 * b + 4 * kq;
      const bgf4 bv4 = *reinterpret_cast<const bgf4 *>(bias + col), gv4 = *reinterpret_cast<const bgf4 *>(epi.gamma + col);
#pragma unroll
      for (int a = 0; a < MI; a++) {
        const int row = m0 + 16 * a + li;
        const bool ok = row < M;
        const size_t rr = ok ? row : M - 1;
        const bgf4 z4 = bz_load4(z2 + rr * epi.ldz + col);
        const float mean_a = epi.stats[2 * rr], rstd_a = epi.stats[2 * rr + 1];
        bgf4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float v = z4[r] + bv4[r], dy = ok ? acc[a][b][r] : 0.f;
          const float sig = tm_sigmoid(v), ah = (v * sig - mean_a) * rstd_a, da = dy * gv4[r];
          const float dact = rstd_a * (da - m1[a] - ah * m2[a]);
          o[r] = ok ? dact * (sig * (1.f + v * (1.f - sig))) : 0.f;
          cg[r] += dy * ah; cb[r] += dy; cz[r] += o[r];
        }
        if (ok) *reinterpret_cast<bgu2 *>(epi.y16 + (size_t)row * epi.ldy16 + col) = bgu2{bg_pack(o.x, o.y), bg_pack(o.z, o.w)};
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { cg[r] = bg_row16_sum(cg[r]); cb[r] = bg_row16_sum(cb[r]); cz[r] = bg_row16_sum(cz[r]); }
      if (li == 0) { *reinterpret_cast<bgf4 *>(pp + col) = cg; *reinterpret_cast<bgf4 *>(pp + BN + col) = cb; *reinterpret_cast<bgf4 *>(pp + 2 * BN + col) = cz; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (EPI == 3 || EPI == 4) {
    // element-wise on the tile (any N: columns beyond N are computed on clamped weight rows and never stored)
    const bool vec = EPI == 3 ? (!(ldc & 3) && !((uintptr_t)C & (4 * sizeof(bz_t) - 1))) : true;
#pragma unroll
    for (int b = 0; b < NI; b++) {
      const int col = n0 + nw + 16 * b + 4 * kq;
      bgf4 bv;
#pragma unroll
      for (int r = 0; r < 4; r++) bv[r] = col + r < N ? bias[col + r] : 0.f;
      bgf4 cz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < MI; a++) {
        const int row = m0 + 16 * a + li;
        const bool ok = row < M;
        bgf4 o;
        if constexpr (EPI == 3) {
#pragma unroll
          for (int r = 0; r < 4; r++) { const float v = acc[a][b][r] + bv[r]; o[r] = tm_silu(v); }
          if (ok) {
            bz_t *zo = reinterpret_cast<bz_t *>(C) + (size_t)row * ldc + col;
            if (vec && col + 3 < N) bz_store4(zo, acc[a][b]);
            else {
#pragma unroll
              for (int r = 0; r < 4; r++) if (col + r < N) zo[r] = bz_from(acc[a][b][r]);
            }
          }
        } else {
          const size_t rr = ok ? row : M - 1;
          bgf4 z4 = {0.f, 0.f, 0.f, 0.f};
          if (!(epi.ldz & 3) && col + 3 < N) z4 = bz_load4(epi.z + rr * epi.ldz + col);       // (z's base is aligned for four elements: checked by the entry point)
          else {
#pragma unroll
            for (int r = 0; r < 4; r++) if (col + r < N) z4[r] = bz_f32(epi.z[rr * epi.ldz + col + r]);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float v = z4[r] + bv[r], sig = tm_sigmoid(v);
            o[r] = ok ? acc[a][b][r] * (sig * (1.f + v * (1.f - sig))) : 0.f;
            cz[r] += o[r];
          }
        }
        if (ok) {
          if (EPI == 3 && epi.yf) {
            float *yo = epi.yf + (size_t)row * epi.ldyf + col;
#pragma unroll
            for (int r = 0; r < 4; r++) if (col + r < N) yo[r] = o[r];
          } else {
            bf16_t *yo = epi.y16 + (size_t)row * epi.ldy16 + col;
            if (col + 3 < N) *reinterpret_cast<bgu2 *>(yo) = bgu2{bg_pack(o.x, o.y), bg_pack(o.z, o.w)};
            else {
#pragma unroll
              for (int r = 0; r < 4; r++) if (col + r < N) yo[r] = (bf16_t)(bg_pack(o[r], 0.f) & 0xffffu);
            }
          }
        }
      }
      if constexpr (EPI == 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) cz[r] = bg_row16_sum(cz[r]);
        if (li == 0) {
          float *pp = epi.partial + (size_t)blockIdx.x * N + col;
#pragma unroll
          for (int r = 0; r < 4; r++) if (col + r < N) pp[r] = cz[r];
        }
      }
    }
  }
#undef li
#undef kq
}

// Unfused backward of a Dense -> SiLU block whose output gradient does not come out of a bf16 GEMM (the value net's last hidden layer: its
// consumer is the 1-wide head on the fp32 kernels): dz = dy silu'(z + bias) as bf16 + per-80-row-tile column sums.  One workgroup per row tile,
// a thread per column (coalesced rows).
__global__ __launch_bounds__(256) void k_bf_silu_bwd(const float *__restrict__ dy, int ldy, const bz_t *__restrict__ z, int ldz, const float *__restrict__ bias,
                                                     bf16_t *__restrict__ dz, int lddz, float *__restrict__ partial, int M, int N) {
  const int r0 = blockIdx.x * 80, r1 = min(M, r0 + 80);
  for (int c = threadIdx.x; c < N; c += 256) {
    const float b = bias[c];
    float sum = 0.f;
    for (int r = r0; r < r1; r++) {
      const float v = bz_f32(z[(size_t)r * ldz + c]) + b, sig = tm_sigmoid(v);
      const float o = dy[(size_t)r * ldy + c] * (sig * (1.f + v * (1.f - sig)));
      dz[(size_t)r * lddz + c] = (bf16_t)(bg_pack(o, 0.f) & 0xffffu);
      sum += o;
    }
    partial[(size_t)blockIdx.x * N + c] = sum;
  }
}

// The same with four consecutive columns per thread (8-byte loads of z, 16-byte loads of dy, 8-byte stores of dz; N a multiple of 4, <= 1024, aligned rows)
// and the row tile's 80 rows dealt to 256 / (N / 4) row phases, whose column sums meet in LDS.  RANK1: the output gradient is the outer product
// dy1[row] w1[col] — the block's consumer is a 1-wide un-activated layer (the value head: d loss / d y = d loss / d baseline x its weight row),
// formed here instead of by an input-gradient GEMM with a contraction length of one that writes [M][N] floats for this kernel to read back.
template <bool RANK1>
__global__ __launch_bounds__(256) void k_bf_silu_bwd4(const float *__restrict__ dy, int ldy, const float *__restrict__ dy1, const float *__restrict__ w1,
                                                      const bz_t *__restrict__ z, int ldz, const float *__restrict__ bias, bf16_t *__restrict__ dz, int lddz,
                                                      float *__restrict__ partial, int M, int N) {
  __shared__ float red[256 * 4];
  const int ncg = N >> 2, nph = 256 / ncg, t = threadIdx.x, cg = t % ncg, ph = t / ncg;
  const int r0 = blockIdx.x * 80, r1 = min(M, r0 + 80);
  if (ph < nph) {
    const bgf4 b = *reinterpret_cast<const bgf4 *>(bias + 4 * cg);
    bgf4 wv = {0.f, 0.f, 0.f, 0.f}, sum = {0.f, 0.f, 0.f, 0.f};
    if (RANK1) wv = *reinterpret_cast<const bgf4 *>(w1 + 4 * cg);
    for (int r = r0 + ph; r < r1; r += nph) {
      const bgf4 zv = bz_load4(z + (size_t)r * ldz + 4 * cg);
      bgf4 d;
      if (RANK1) { const float g = dy1[r]; d = bgf4{g * wv.x, g * wv.y, g * wv.z, g * wv.w}; }
      else d = *reinterpret_cast<const bgf4 *>(dy + (size_t)r * ldy + 4 * cg);
      bgf4 o;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float v = zv[j] + b[j], sig = tm_sigmoid(v);
        o[j] = d[j] * (sig * (1.f + v * (1.f - sig)));
      }
      *reinterpret_cast<bgu2 *>(dz + (size_t)r * lddz + 4 * cg) = bgu2{bg_pack(o.x, o.y), bg_pack(o.z, o.w)};
      sum += o;
    }
    *reinterpret_cast<bgf4 *>(red + 4 * t) = sum;
  }
  __syncthreads();
  for (int c = t; c < N; c += 256) {
    float v = 0.f;
    for (int q = 0; q < nph; q++) v += red[4 * (q * ncg + (c >> 2)) + (c & 3)];
    partial[(size_t)blockIdx.x * N + c] = v;
  }
}

// ---- dW = dY^T X ----------------------------------------------------------------------------------------------------------------------
// LDS image of a [rows][128 bf16] tile (256-byte rows) for transposed reads: byte offset of 16-byte chunk ch (0 .. 15) of row r; the XOR
// makes the two 4-row blocks a 32-lane half reads (8 rows apart, same columns) land on disjoint banks
__device__ __forceinline__ int bg_tr_off(int r, int ch) { return 256 * r + 16 * (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))); }

#define BGDW_BT 128      // output tile 128 (n) x 128 (k)
#define BGDW_BM 32       // rows of M per LDS stage = one MFMA k step
#define BGDW_STAGE (2 * BGDW_BM * 256)
// 4 waves, each 64 (n) x 64 (k) of the tile = 4 x 4 MFMA tiles.  YF32 / XF32: the operand is fp32 in memory (converted at the LDS write).
template <bool YF32, bool XF32>
__device__ __forceinline__ void bgemm_dw_tile(const void *__restrict__ dYv, int ldy, const void *__restrict__ Xv, int ldx, float *__restrict__ slabs,
                                              int M, int N, int K, int with_bias, int rows_per_split, int ld_slab, int tile_n, int tile_k, int split) {
  extern __shared__ __attribute__((aligned(16))) char bg_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, i16 = lane & 15;
  const int n0 = tile_n * BGDW_BT, k0 = tile_k * BGDW_BT;
  const int r_begin = split * rows_per_split, r_end = min(M, r_begin + rows_per_split);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  // staging in units of 16 bytes of memory (whole contiguous row segments per wave instruction, see k_bgemm_nt): fp32 operand: unit = 4
  // columns -> 8 bytes of LDS, 32 units per row, 4 passes of 256 threads per 32-row stage; bf16: unit = 8 columns -> 16 bytes, 16 per row, 2 passes.
  // Columns beyond the matrix: a unit that starts inside the row's allocation (ld) is loaded whole and masked to zero beyond the width; rows
  // beyond r_end must be exact zeros (they are summed over): not loaded.
  constexpr int YU = YF32 ? 32 : 16, XU = XF32 ? 32 : 16, YP = YU / 8, XP = XU / 8, YE = 128 / YU, XE = 128 / XU;
  struct Stage { bgu4 y[YP], x[XP]; };
  bgf4 colacc[2];                          // sums over this thread's rows of its dY columns (bias gradient): 4 resp. 8 columns
  colacc[0] = bgf4{0.f, 0.f, 0.f, 0.f};
  colacc[1] = bgf4{0.f, 0.f, 0.f, 0.f};
  const int yu = t % YU, xu = t % XU;       // (256 % YU == 0: the unit of a thread is the same in every pass)
  auto ldu = [&](const void *base, int ld, int row, int col, int width, bool f32, bool ok) -> bgu4 {
    bgu4 raw = {0u, 0u, 0u, 0u};
    if (ok && col < ld) {
      raw = f32 ? *reinterpret_cast<const bgu4 *>(reinterpret_cast<const float *>(base) + (size_t)row * ld + col)
                : *reinterpret_cast<const bgu4 *>(reinterpret_cast<const bf16_t *>(base) + (size_t)row * ld + col);
      if (col + (f32 ? 4 : 8) > width) {
        if (f32) {
#pragma unroll
          for (int j = 0; j < 4; j++) if (col + j >= width) raw[j] = 0u;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (col + 2 * j >= width) raw[j] = 0u;
            else if (col + 2 * j + 1 >= width) raw[j] &= 0xffffu;
          }
        }
      }
    }
    return raw;
  };
  auto gload = [&](Stage &R, int r0) {
#pragma unroll
    for (int p = 0; p < YP; p++) {
      const int row = r0 + (t + 256 * p) / YU;
      R.y[p] = ldu(dYv, ldy, min(row, M - 1), n0 + YE * yu, N, YF32, row < r_end);
    }
#pragma unroll
    for (int p = 0; p < XP; p++) {
      const int row = r0 + (t + 256 * p) / XU;
      R.x[p] = ldu(Xv, ldx, min(row, M - 1), k0 + XE * xu, K, XF32, row < r_end);
    }
  };
  auto swrite = [&](const Stage &R, int stage) {
    char *sy = bg_lds + stage * BGDW_STAGE, *sx = sy + BGDW_BM * 256;
#pragma unroll
    for (int p = 0; p < YP; p++) {
      const int r = (t + 256 * p) / YU;
      if (YF32) {
        const bgf4 v = __builtin_bit_cast(bgf4, R.y[p]);
        colacc[0] += v;
        *reinterpret_cast<bgu2 *>(sy + bg_tr_off(r, yu >> 1) + 8 * (yu & 1)) = bgu2{bg_pack(v.x, v.y), bg_pack(v.z, v.w)};
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          colacc[0][j] += bg_f32((bf16_t)(R.y[p][j >> 1] >> (16 * (j & 1))));
          colacc[1][j] += bg_f32((bf16_t)(R.y[p][2 + (j >> 1)] >> (16 * (j & 1))));
        }
        *reinterpret_cast<bgu4 *>(sy + bg_tr_off(r, yu)) = R.y[p];
      }
    }
#pragma unroll
    for (int p = 0; p < XP; p++) {
      const int r = (t + 256 * p) / XU;
      if (XF32) {
        const bgf4 v = __builtin_bit_cast(bgf4, R.x[p]);
        *reinterpret_cast<bgu2 *>(sx + bg_tr_off(r, xu >> 1) + 8 * (xu & 1)) = bgu2{bg_pack(v.x, v.y), bg_pack(v.z, v.w)};
      } else {
        *reinterpret_cast<bgu4 *>(sx + bg_tr_off(r, xu)) = R.x[p];
      }
    }
  };
  // transposed reads: lane (group g, q = (lane & 15) >> 2, p = lane & 3) supplies the address of row 8 g (+ 4) + q, columns 16 tile + 4 p .. + 3;
  // it receives, for column (lane & 15) of the block, the four rows.  Two reads = the eight k of the lane's MFMA operand.
  const int q4 = i16 >> 2, p4 = i16 & 3;
  auto tr_addr = [&](int row, int col16) { return bg_tr_off(row, 2 * col16 + (p4 >> 1)) + 8 * (p4 & 1); };
  bgf4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = bgf4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int stage) {
    const char *sy = bg_lds + stage * BGDW_STAGE, *sx = sy + BGDW_BM * 256;
    bgs8 fy[4], fx[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const bgs4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bgs4 *)BG_LDS(sy + tr_addr(8 * g + q4, wn / 16 + a)));
      const bgs4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bgs4 *)BG_LDS(sy + tr_addr(8 * g + 4 + q4, wn / 16 + a)));
      fy[a] = bgs8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const bgs4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bgs4 *)BG_LDS(sx + tr_addr(8 * g + q4, wk / 16 + b)));
      const bgs4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bgs4 *)BG_LDS(sx + tr_addr(8 * g + 4 + q4, wk / 16 + b)));
      fx[b] = bgs8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
    // X fragment first: the tile comes out as D[k][n], i.e. a lane holds four consecutive k of one output row n (one dwordx4 per tile)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bgb8, fx[b]), __builtin_bit_cast(bgb8, fy[a]), acc[a][b], 0, 0, 0);
  };
  const int nt = (r_end - r_begin + BGDW_BM - 1) / BGDW_BM;
  if (nt > 0) {
    Stage R;
    gload(R, r_begin);
    swrite(R, 0);
    __syncthreads();
    for (int it = 0; it < nt; it++) {
      const int cur = it & 1;
      gload(R, r_begin + (it + 1) * BGDW_BM);          // (past r_end: all rows masked -> zeros, written to the stage nobody reads)
      compute(cur);
      swrite(R, cur ^ 1);
      __syncthreads();
    }
  }
  float *out = slabs + (size_t)split * (size_t)N * ld_slab;
  if (with_bias && tile_k == 0) {            // thread t holds the sums of dY columns n0 + YE yu .. of the rows t / YU + (256 / YU) j
    __syncthreads();
    float *red = reinterpret_cast<float *>(bg_lds);            // [256 / YU][128]
    *reinterpret_cast<bgf4 *>(red + (t / YU) * BGDW_BT + YE * yu) = colacc[0];
    if (!YF32) *reinterpret_cast<bgf4 *>(red + (t / YU) * BGDW_BT + YE * yu + 4) = colacc[1];
    __syncthreads();
    if (t < BGDW_BT) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 256 / YU; j++) v += red[j * BGDW_BT + t];
      if (n0 + t < N) out[(size_t)(n0 + t) * ld_slab + K] = v;
    }
  }
  // register r of tile (a, b): output row n0 + wn + 16 a + (lane & 15), columns k0 + wk + 16 b + 4 (lane >> 4) + r
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int row = n0 + wn + 16 * a + i16;
    if (row < N) {
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int col = k0 + wk + 16 * b + 4 * g;
        float *o = out + (size_t)row * ld_slab + col;
        if (col + 3 < K) *reinterpret_cast<bgf4 *>(o) = acc[a][b];
        else {
#pragma unroll
          for (int r = 0; r < 4; r++) if (col + r < K) o[r] = acc[a][b][r];
        }
      }
    }
  }
}
// 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so all tiles of one row slab — which re-read the
// same rows of dY and X — are given ids of ONE residue class mod 8: id = 8 q + xcd -> slab (q / tiles) 8 + xcd, tile q % tiles.
template <bool YF32, bool XF32>
__global__ __launch_bounds__(256) void k_bgemm_dw(const void *__restrict__ dY, int ldy, const void *__restrict__ X, int ldx, float *__restrict__ slabs,
                                                  int M, int N, int K, int with_bias, int rows_per_split, int ld_slab, int tiles_n, int tiles_k, int S) {
  const int tiles = tiles_n * tiles_k, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int split = (q / tiles) * 8 + xcd, tl = q % tiles;
  if (split >= S) return;
  bgemm_dw_tile<YF32, XF32>(dY, ldy, X, ldx, slabs, M, N, K, with_bias, rows_per_split, ld_slab, tl / tiles_k, tl % tiles_k, split);
}
// dW[n][k] = sum over the slabs; db[n] = the slabs' column K.  One lane per output element, eight loads in flight.
__global__ __launch_bounds__(256) void k_bgemm_dw_reduce(const float *__restrict__ slabs, float *__restrict__ dW, float *__restrict__ db, int S, int N, int K,
                                                         int with_bias, int ld_slab, int lddw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int Kx = K + (with_bias ? 1 : 0);
  if (i >= (long long)N * Kx) return;
  const int n = (int)(i / Kx), k = (int)(i % Kx);
  const float *p = slabs + (size_t)n * ld_slab + k;
  const size_t step = (size_t)N * ld_slab;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= S; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] += p[(size_t)(s + u) * step];
  }
  for (; s < S; s++) a[0] += p[(size_t)s * step];
  const float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (k < K) dW[(size_t)n * lddw + k] = v; else db[n] = v;
}


// ---- every weight gradient of a backward pass in ONE launch (+ one reduction launch): the grouped form of csrc/gemm_kernels.h (k_gemm_dw_grouped) for
// the bf16 operands.  Launched layer by layer a weight gradient has to split its rows into ~32 slabs to fill the chip on its own — 32 fp32 copies of the
// weight matrix written and read back per layer; side by side the problems of a backward pass fill it with 4 - 8 slabs each.  The table travels by value
// (hipGraph-safe); a problem's workgroups keep the XCD-aware order of k_bgemm_dw (all tiles of one row slab in one residue class mod 8).
#define BDW_GROUP_MAX 24
struct BdwProblem {
  const void *dY, *X;
  float *dW, *db, *slabs;
  int ldy, ldx, lddw, M, N, K, y_f32, x_f32, rows_per_split, S, ld_slab, tiles_n, tiles_k;
  int wg_begin, red_begin;
};
struct BdwGroup { BdwProblem p[BDW_GROUP_MAX]; int n; };
__global__ __launch_bounds__(256) void k_bgemm_dw_grouped(const BdwGroup G) {
  int i = 0;
  for (int j = 1; j < G.n; j++) if ((int)blockIdx.x >= G.p[j].wg_begin) i = j;       // uniform
  const BdwProblem &P = G.p[i];
  const int local = blockIdx.x - P.wg_begin, tiles = P.tiles_n * P.tiles_k, xcd = local & 7, q = local >> 3;
  const int split = (q / tiles) * 8 + xcd, tl = q % tiles;
  if (split >= P.S) return;
  const int wb = P.db != nullptr;
  if (P.y_f32 && P.x_f32) bgemm_dw_tile<true, true>(P.dY, P.ldy, P.X, P.ldx, P.slabs, P.M, P.N, P.K, wb, P.rows_per_split, P.ld_slab, tl / P.tiles_k, tl % P.tiles_k, split);
  else if (P.y_f32) bgemm_dw_tile<true, false>(P.dY, P.ldy, P.X, P.ldx, P.slabs, P.M, P.N, P.K, wb, P.rows_per_split, P.ld_slab, tl / P.tiles_k, tl % P.tiles_k, split);
  else if (P.x_f32) bgemm_dw_tile<false, true>(P.dY, P.ldy, P.X, P.ldx, P.slabs, P.M, P.N, P.K, wb, P.rows_per_split, P.ld_slab, tl / P.tiles_k, tl % P.tiles_k, split);
  else bgemm_dw_tile<false, false>(P.dY, P.ldy, P.X, P.ldx, P.slabs, P.M, P.N, P.K, wb, P.rows_per_split, P.ld_slab, tl / P.tiles_k, tl % P.tiles_k, split);
}
__global__ __launch_bounds__(256) void k_bgemm_dw_reduce_grouped(const BdwGroup G) {
  int i = 0;
  for (int j = 1; j < G.n; j++) if ((int)blockIdx.x >= G.p[j].red_begin) i = j;
  const BdwProblem &P = G.p[i];
  const int wb = P.db != nullptr;
  const long long e = (long long)(blockIdx.x - P.red_begin) * 256 + threadIdx.x;
  const int Kx = P.K + wb;
  if (e >= (long long)P.N * Kx) return;
  const int n = (int)(e / Kx), k = (int)(e % Kx);
  const float *p = P.slabs + (size_t)n * P.ld_slab + k;
  const size_t step = (size_t)P.N * P.ld_slab;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= P.S; s += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] += p[(size_t)(s + u) * step];
  }
  for (; s < P.S; s++) a[0] += p[(size_t)s * step];
  const float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (k < P.K) P.dW[(size_t)n * P.lddw + k] = v; else P.db[n] = v;
}


// ---- The ACTING policy's dense layer in bf16 GEMM-input mode, without LDS (the bf16 twin of csrc/ppo_kernels.h: k_linear_nolds_mfma; the acting
// policy runs next to the other env groups' physics kernel, which owns every CU's LDS).  C[M][N] = A[M][K] W^T (+ bias): the fp32 activations are
// rounded to bf16 in registers (v_cvt_pk_bf16_f32 — the rounding the learner's kernels apply when they stage a tile), W is the layer's resident
// bf16 shadow [N][ldw] (zero padded to a multiple of 64 columns: tmjx_bf16_shadow), fp32 accumulate on v_mfma_f32_16x16x32_bf16: one
// instruction per 32 k where the fp32 kernel needs eight, half the weight bytes — and the SAME operand rounding as the SGD forward pass, so that
// the behaviour log-prob of a roll-out and the learner's first-pass log-prob agree (the PPO ratio starts at 1).
// One wave per 32 x 32 output tile; a lane's 8 consecutive k (k = 32 s + 8 (lane / 16) ..) feed one operand register of each of the 2 x 2 tiles.
// `mean` / `inv_std` (optional, [K]): the operand is (A - mean) * inv_std (the observation normaliser, first layer).
template <bool A_KMAJOR>
__global__ __launch_bounds__(64, 5) void k_linear_nolds_bf16(const float *__restrict__ A, long long sa_row, long long sa_k, const bf16_t *__restrict__ W, int ldw,
                                                          const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                          const float *__restrict__ mean, const float *__restrict__ inv_std) {
  TM_PRIO_ACTING_BF();
  const int lane = threadIdx.x, li = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
  int ar[2], wc[2];
#pragma unroll
  for (int t = 0; t < 2; t++) { int r = row0 + 16 * t + li; ar[t] = r < M ? r : M - 1; int c = col0 + 16 * t + li; wc[t] = c < N ? c : N - 1; }
  bgf4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = bgf4{0.f, 0.f, 0.f, 0.f};
  struct Frag { bgu4 a[2], w[2]; };
  auto load = [&](Frag &f, int k0) {
    const int k = k0 + 8 * kq;                   // K is a multiple of 4 (host check); the shadow's row is zero beyond K up to ldw (a multiple of 64)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      bgf4 v[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int kh = k + 4 * h;
        const bool ok = kh < K;                  // a float4 of A is inside or outside as a whole
        const int kc = ok ? kh : 0;
        bgf4 x;
        if (A_KMAJOR) {
          x.x = A[(long long)kc * sa_k + ar[t]]; x.y = A[(long long)(kc + 1) * sa_k + ar[t]];
          x.z = A[(long long)(kc + 2) * sa_k + ar[t]]; x.w = A[(long long)(kc + 3) * sa_k + ar[t]];
        } else {
          x = *reinterpret_cast<const bgf4 *>(A + (long long)ar[t] * sa_row + kc);
        }
        if (mean) {
          const bgf4 mu = *reinterpret_cast<const bgf4 *>(mean + kc), is = *reinterpret_cast<const bgf4 *>(inv_std + kc);
          x = (x - mu) * is;
        }
        v[h] = ok ? x : bgf4{0.f, 0.f, 0.f, 0.f};
      }
      f.a[t] = bg_pack8(v[0], v[1]);
      f.w[t] = *reinterpret_cast<const bgu4 *>(W + (size_t)wc[t] * ldw + (k < ldw ? k : 0));      // (k < ldw always holds while k0 < K)
    }
  };
  auto mma = [&](const Frag &f) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bgb8, f.a[a]), __builtin_bit_cast(bgb8, f.w[b]), acc[a][b], 0, 0, 0);
  };
  // TWO K steps in flight (a step is ONE matrix instruction per tile: what bounds a wave is the latency of its operand loads).  Round 4 kept three —
  // at 144 VGPRs + 48 AGPRs, and three physics waves of 136 allocated registers leave 104 of a SIMD's 512: such a wave could only start where a
  // physics wave had retired.  Two steps fit __launch_bounds__(64, 5)'s 96 registers without a spill; same instructions in the same k order.
  Frag f0, f1;
  load(f0, 0);
  for (int k0 = 0; k0 < K; k0 += 64) {
    if (k0 + 32 < K) load(f1, k0 + 32);
    mma(f0);
    if (k0 + 32 >= K) break;
    if (k0 + 64 < K) load(f0, k0 + 64);
    mma(f1);
  }
  // accumulator component r of lane l holds C[4 (l / 16) + r][l % 16] of its 16 x 16 tile
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int c = col0 + 16 * b + li;
      const float bv = (bias && c < N) ? bias[c] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + 16 * a + 4 * kq + r;
        if (row < M && c < N) C[(size_t)row * N + c] = acc[a][b][r] + bv;
      }
    }
}
