// csrc/gemm_kernels.h — the dense contractions of the PPO learner (K4 / K5 / K8) on the matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf chain, MI355X_MICROARCH.md), written for the shapes this path has:
// a tall activation matrix (rows = unroll_length x minibatch rows = 20 480 per GPU, 40 960 in BASELINE config 5) against small
// weight matrices (64 .. 1024 wide).  Reference layers: flax nn.Dense in track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76
// and brax's value MLP (ppo_networks.py:180-184); their gradients are what jax.grad derives for losses.py:103-245.
//
//   k_gemm_act<NIW, BT>   C[M][N] = A[M][K] . op(W) (+ bias):  BT = true  W is [N][K] (y = x W^T + b, the forward pass)
//                                                              BT = false W is [K][N] (dx = dy W, the input gradient)
//       Workgroup tile 80 rows x 64 NIW columns, 4 waves side by side along N, each 5 x NIW tiles of 16 x 16.  80 rows because
//       20 480 = 256 x 80: one workgroup per CU, no tail wave (a 128-row tile leaves 320 workgroups on 256 CUs: 62 % of the chip).
//       K is walked in steps of 32 through two LDS stages; the next stage's global loads are issued before this stage's MFMAs
//       and written to LDS behind them (register staging: one barrier per step).  A lane's float4 along K feeds four MFMAs (any
//       assignment of the k of a step to the MFMAs is valid as long as both operands use the same one), so the activation
//       fragments — and for BT the weight fragments — are ds_read_b128; rows are padded to 40 floats, which makes every 16-lane
//       group of a b128 read hit 16 distinct 16-byte slots (row stride 10 slots: rows of the even-kq lanes land on even slots, of
//       the odd-kq lanes on odd slots).  BT = false reads the weight fragment as ds_read_b32 from a [k][n] image whose row stride
//       is 4 mod 8 floats (lanes of kq and kq + 1 then sit 16 banks apart).
//   k_gemm_dw             dW[N][K] = dY[M][N]^T . X[M][K] and db[N] = column sums of dY, as slabs over row ranges of M
//       (the contraction runs over the 20 480 rows: 4 .. 24 output tiles of 128 x 128 would leave the chip idle, so the rows are
//       split until there are about 256 workgroups) + k_dw_reduce.  The bias gradient rides along: every thread adds up the dY
//       values it stages (its four columns, all its rows), the eight partial sums per column meet in LDS after the loop and land in
//       column K of the slab: no separate column-sum launches.
#pragma once
#include "silu_math.h"
#include <type_traits>

typedef float __attribute__((ext_vector_type(4))) gf4;
typedef float __attribute__((ext_vector_type(2))) gf2;

#define GEMM_BM 80
#define GEMM_BK 32
#define GEMM_A_ROWS (GEMM_BM + 8)   // LDS rows of the A image: the tile + 8 dump rows (see k_gemm_act)
#define GEMM_LDA 40          // floats per LDS row of a [rows][k] image (32 + 8 pad)

// Four consecutive elements k .. k + 3 of one matrix row, zero beyond `K` and for rows outside the matrix.  BRANCH-FREE on purpose:
// every load is issued unconditionally from a clamped (valid) address and masked afterwards, so the compiler can leave the loads
// of the next K step in flight across this step's MFMAs (a load inside a divergent branch is waited for at the end of the branch:
// the first version of these kernels ran at 40 % of the MFMA rate for exactly that reason).
//   VEC  : the row base is 16-byte aligned AND the row's allocation reaches the end of the last float4 (ld >= K rounded up to 4):
//          one dwordx4 load, elements beyond K zeroed;  otherwise four dword loads.
// The masking is a SEPARATE step (gemm_mask4, applied when the staged registers are written to LDS, behind the MFMAs of the current
// step): applied right at the load it makes the loaded value live — and waited for — before the MFMAs start.
template <bool VEC>
__device__ __forceinline__ gf4 gemm_ld4(const float *__restrict__ p, long long row_off, int k, int K, bool row_ok, int &mask) {
  const float *q = p + (row_ok ? row_off : 0ll);
  const bool any = row_ok && k < K;
  const int left = any ? K - k : 0;                    // valid elements from k on
  mask = left >= 4 ? 15 : (1 << (left > 0 ? left : 0)) - 1;
  gf4 v;
  if (VEC) {                                  // compile time: a run-time flag here puts every load behind a branch and its vmcnt(0)
    v = *reinterpret_cast<const gf4 *>(q + (any ? k : 0));
  } else {
    const int last = K - 1;
    v.x = q[any ? k : 0];
    v.y = q[any ? min(k + 1, last) : 0];
    v.z = q[any ? min(k + 2, last) : 0];
    v.w = q[any ? min(k + 3, last) : 0];
  }
  return v;
}
__device__ __forceinline__ gf4 gemm_mask4(gf4 v, int mask) {
  return gf4{(mask & 1) ? v.x : 0.f, (mask & 2) ? v.y : 0.f, (mask & 4) ? v.z : 0.f, (mask & 8) ? v.w : 0.f};
}

// LLVM SchedGroupMask values for __builtin_amdgcn_sched_group_barrier
#define GEMM_SGB(a, b, c) __builtin_amdgcn_sched_group_barrier(a, b, c)
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_VMEM_READ 0x20
#define SG_VMEM_WRITE 0x40
#define SG_DS_READ 0x100
#define SG_DS_WRITE 0x200

// Fused forward epilogue of a Dense -> SiLU -> LayerNorm block (LN = true; the tile must span the whole row: N == 64 NIW, one column
// block): z = A W^T goes to C as before (WITHOUT the bias: what k_silu_ln_bwd expects), y = LayerNorm(silu(z + bias)) to `y`, the row's
// (mean, 1 / std) to `stats` — the arithmetic of k_silu_ln_fwd (two-pass variance), without its launch and its re-read of z.
struct GemmLN {
  const float *gamma, *beta; float *y, *stats; float eps;      // EPI 1 (forward);  EPI 2 reads gamma and stats
  const float *z; float *partial;                              // EPI 2: the block's saved pre-activation z; per-workgroup column sums [3][BN]
};
// EPI 4 (input-gradient kernel, any tile): the SiLU backward of the producing Dense -> SiLU layer on the accumulators (see the epilogue).
// EPI 2 (input-gradient kernel, tile = whole rows, N == 256): the tile is d loss / d y of a Dense -> SiLU -> LayerNorm block (y its output);
// the epilogue applies that block's LayerNorm + SiLU backward (the arithmetic of k_silu_ln_bwd) and stores d loss / d z instead, plus the
// workgroup's column sums of (dy * ahat | dy | dz) = partial d gamma | d beta | d bias: the block's own backward launch disappears.
// sum over the 16 lanes of a DPP row (all 16 lanes get it): quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float gemm_row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, false));
  return x;
}

template <int NIW> struct GemmCfg { static constexpr int NWAVE = NIW >= 2 ? 8 : 4, THREADS = 64 * NWAVE, NI = 4 * NIW / NWAVE; };
// MT: 16-row tiles per workgroup tile — 5 (80 rows: one workgroup per CU at 20 480 rows) or 2 (32 rows: 5 120-row problems, one rank's share of the
// 8-GPU configuration, are 160 workgroups instead of 64 on the 256 CUs; tmjx_hip.hip: gemm_mt)
template <int NIW, bool BT, bool AVEC, bool WVEC, int EPI = 0, int MT = 5>
__global__ __launch_bounds__(GemmCfg<NIW>::THREADS) void k_gemm_act(const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw, const float *__restrict__ bias,
                                                  float *__restrict__ C, int ldc, int M, int N, int K, GemmLN ln = GemmLN{}) {
  constexpr int BN = 64 * NIW;
  constexpr int BM = 16 * MT, A_ROWS = BM + 8;      // rows of the workgroup tile; LDS rows of the A image (the tile + 8 dump rows)
  constexpr int LDB_T = GEMM_LDA;            // BT: [n][k] image, 40 floats per row
  constexpr int LDB_N = BN + 4;              // !BT: [k][n] image, row stride 4 mod 8
  // (8 extra rows behind the A tile: the lanes of the partly empty last staging pass write THERE instead of sitting out a divergent
  // branch — a branch in the K step splits its basic block, which voids the sched_group_barrier interleave and costs a waitcnt)
  constexpr int A_FLOATS = A_ROWS * GEMM_LDA, B_FLOATS = BT ? BN * LDB_T : GEMM_BK * LDB_N, STAGE = A_FLOATS + B_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];
  // TWO waves per SIMD (8 per workgroup) once the tile is 128 columns or wider: with one wave per SIMD everything that is not an
  // MFMA (address arithmetic, masks, LDS traffic, the barrier) idles the matrix pipe — measured 52 % MFMA-busy; the second wave's
  // MFMAs fill those gaps.  The waves sit side by side along N: 16 NI columns each.
  constexpr int NT = GemmCfg<NIW>::THREADS, NI = GemmCfg<NIW>::NI;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, nw = wave * 16 * NI;      // this wave's first column inside the tile
  // ---- global -> register staging (float4 per thread): A 80 x 8 float4 (the last pass partly empty), B BN x 8 or 32 x BN / 4
  constexpr int A_PASS = (BM * 8 + NT - 1) / NT, B_PASS = BT ? (BN * 8) / NT : (GEMM_BK * (BN / 4)) / NT, NPASS = A_PASS + B_PASS;
  // TWO register sets: tile j travels in set j & 1, loaded a whole K step before it is written to LDS (the write sits between the
  // MFMAs of the step in between, so nothing waits for memory)
  struct Stage { gf4 v[NPASS]; int m[NPASS]; };
  auto gload = [&](Stage &R, int k0) {
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + NT * p, r = f >> 3, c4 = f & 7;
      const bool ok = f < BM * 8 && m0 + r < M;
      R.v[p] = gemm_ld4<AVEC>(A, (long long)(m0 + (ok ? r : 0)) * lda, k0 + 4 * c4, K, ok, R.m[p]);
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++) {
      const int f = t + NT * p;
      if (BT) {
        const int r = f >> 3, c4 = f & 7;
        const bool ok = n0 + r < N;
        R.v[A_PASS + p] = gemm_ld4<WVEC>(W, (long long)(n0 + (ok ? r : 0)) * ldw, k0 + 4 * c4, K, ok, R.m[A_PASS + p]);
      } else {
        const int r = f / (BN / 4), c4 = f % (BN / 4);               // row = k, four consecutive output columns
        const bool ok = k0 + r < K;
        R.v[A_PASS + p] = gemm_ld4<WVEC>(W, (long long)(ok ? k0 + r : 0) * ldw + n0, 4 * c4, N - n0, ok, R.m[A_PASS + p]);
      }
    }
  };
  auto swrite = [&](const Stage &R, int stage) {
    float *sa = gemm_lds + stage * STAGE, *sb = sa + A_FLOATS;
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + NT * p, r = f < BM * 8 ? f >> 3 : BM + ((f >> 3) & 7), c4 = f & 7;
      *reinterpret_cast<gf4 *>(sa + r * GEMM_LDA + 4 * c4) = gemm_mask4(R.v[p], R.m[p]);
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++) {
      const int f = t + NT * p;
      const gf4 v = gemm_mask4(R.v[A_PASS + p], R.m[A_PASS + p]);
      if (BT) { const int r = f >> 3, c4 = f & 7; *reinterpret_cast<gf4 *>(sb + r * LDB_T + 4 * c4) = v; }
      else { const int r = f / (BN / 4), c4 = f % (BN / 4); *reinterpret_cast<gf4 *>(sb + r * LDB_N + 4 * c4) = v; }
    }
  };
  // FAST forms for the tiles that lie entirely inside K (all but the last one or two): no masks at all.  Rows / columns outside the
  // matrix are CLAMPED to valid ones instead of zeroed — row m of C depends only on row m of A and column n only on row / column n of W,
  // and rows >= M, columns >= N are never stored — so the only elements that must be exact zeros are those with k >= K, and those exist only
  // in the last (partial) tile.  Loop-invariant per-pass pointers; the K offset is wave-uniform (scalar).  Measured: the address + mask
  // arithmetic of the masked form (22 vector instructions per load) cost 14 % of the MFMA rate.
  constexpr bool FAST = AVEC && WVEC;
  const float *pa[A_PASS], *pb[B_PASS];
#pragma unroll
  for (int p = 0; p < A_PASS; p++) {
    const int f = t + NT * p, r = f < BM * 8 ? f >> 3 : 0, c4 = f & 7;
    pa[p] = A + (long long)min(m0 + r, M - 1) * lda + 4 * c4;
  }
#pragma unroll
  for (int p = 0; p < B_PASS; p++) {
    const int f = t + NT * p;
    if (BT) { const int r = f >> 3, c4 = f & 7; pb[p] = W + (long long)min(n0 + r, N - 1) * ldw + 4 * c4; }
    else { const int r = f / (BN / 4), c4 = f % (BN / 4); pb[p] = W + (long long)r * ldw + min(n0 + 4 * c4, ((N + 3) & ~3) - 4); }
  }
  auto gload_fast = [&](Stage &R, int k0) {
    const int k0c = k0 + GEMM_BK <= K ? k0 : 0;        // tiles past the end are never consumed: any valid address will do
#pragma unroll
    for (int p = 0; p < A_PASS; p++) R.v[p] = *reinterpret_cast<const gf4 *>(pa[p] + k0c);
#pragma unroll
    for (int p = 0; p < B_PASS; p++) R.v[A_PASS + p] = *reinterpret_cast<const gf4 *>(pb[p] + (BT ? (long long)k0c : (long long)k0c * ldw));
  };
  auto swrite_fast = [&](const Stage &R, int stage) {
    float *sa = gemm_lds + stage * STAGE, *sb = sa + A_FLOATS;
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + NT * p, r = f < BM * 8 ? f >> 3 : BM + ((f >> 3) & 7), c4 = f & 7;
      *reinterpret_cast<gf4 *>(sa + r * GEMM_LDA + 4 * c4) = R.v[p];
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++) {
      const int f = t + NT * p;
      if (BT) { const int r = f >> 3, c4 = f & 7; *reinterpret_cast<gf4 *>(sb + r * LDB_T + 4 * c4) = R.v[A_PASS + p]; }
      else { const int r = f / (BN / 4), c4 = f % (BN / 4); *reinterpret_cast<gf4 *>(sb + r * LDB_N + 4 * c4) = R.v[A_PASS + p]; }
    }
  };
  struct Frag { gf4 a[MT], b[NI]; };
  auto fread = [&](Frag &F, int stage, int c) {
    const float *sa = gemm_lds + stage * STAGE, *sb = sa + A_FLOATS;
#pragma unroll
    for (int a = 0; a < MT; a++) F.a[a] = *reinterpret_cast<const gf4 *>(sa + (16 * a + li) * GEMM_LDA + 16 * c + 4 * kq);
    if (BT) {
#pragma unroll
      for (int b = 0; b < NI; b++) F.b[b] = *reinterpret_cast<const gf4 *>(sb + (nw + 16 * b + li) * LDB_T + 16 * c + 4 * kq);
    } else {
#pragma unroll
      // [k][n] image: lane (li, kq) needs W[k = 16 c + 4 kq + e][its column of tile b].  The tile's columns are INTERLEAVED — tile b holds
      // columns nw + NI li + b — so that the NI tiles' values of one k are NI consecutive floats: one ds_read_b64 per k instead of NI
      // ds_read_b32 (9 instead of 13 LDS reads per chunk; the input-gradient kernel ran 20 % behind the forward one for exactly these reads),
      // and the epilogue stores NI consecutive floats per lane.
      for (int e = 0; e < 4; e++) {
        const float *q = sb + (16 * c + 4 * kq + e) * LDB_N + nw + NI * li;
        if constexpr (NI == 2) { const gf2 v = *reinterpret_cast<const gf2 *>(q); F.b[0][e] = v.x; F.b[1][e] = v.y; }
        else {
#pragma unroll
          for (int b = 0; b < NI; b++) F.b[b][e] = q[b];
        }
      }
    }
  };
  gf4 acc[MT][NI];
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < NI; b++) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const Frag &F) {
    // e outermost: 5 NI independent accumulators between two uses of the same one (dependent latency 40 cycles > issue 32)
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NI; b++) {
          // BT: the weight fragment goes in as the FIRST operand, i.e. the tile comes out transposed: register r of lane (li, kq) is
          // C[16 a + li][16 b + 4 kq + r] — four consecutive columns of one row, stored as one dwordx4 (10 stores per wave instead of 40)
          // (!BT with two interleaved tiles, see fread: likewise — the lane then holds EIGHT consecutive columns 8 kq .. 8 kq + 7 of a row)
          if constexpr (BT || NI == 2) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.b[b][e], F.a[a][e], acc[a][b], 0, 0, 0);
          else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[a][e], F.b[b][e], acc[a][b], 0, 0, 0);
        }
  };
  constexpr int NMFMA = 4 * MT * NI;                       // MFMAs per 16-deep chunk and wave
  constexpr int NFR = MT + (BT ? NI : 4 * NI);          // LDS reads per chunk
  // One K step (32 deep = two chunks of 16) of tile k in stage s, ONE barrier in its middle and MFMAs on both sides of it:
  //   phase A: chunk 0's MFMAs, between them the fragment reads of chunk 1 and the LDS writes of tile k + 1 (register set RW, in
  //            flight since a step and a half) into the other stage — last read before the previous barrier;
  //   barrier: the other stage is complete;
  //   phase B: chunk 1's MFMAs (fragments already in registers: no LDS latency behind the barrier), between them the first
  //            fragments of tile k + 1 and the global loads of tile k + 3 into RW's registers (free again).
  // Branch-free: past the last tile the loads are fully masked and the writes go to a stage nobody reads any more.
#ifndef GEMM_LOAD_VALU
#define GEMM_LOAD_VALU 16   // vector instructions (address + mask arithmetic) the scheduler may place in front of each pinned global load
#endif
#ifndef GEMM_ABL
#define GEMM_ABL 0          // timing ablations (wrong results): 1 no global loads, 2 no LDS writes, 4 no barrier, 8 no fragment reads
#endif
  constexpr int SPREAD = (NMFMA - NFR) / (NPASS + 1) > 0 ? (NMFMA - NFR) / (NPASS + 1) : 1;      // MFMAs between two pinned LDS writes / global loads
  auto kstep = [&](auto fast_tag, Frag &F0, Frag &F1, Stage &RW, int stage, int k_next3) {
    constexpr bool F = decltype(fast_tag)::value;
    if (!(GEMM_ABL & 8)) fread(F1, stage, 1);
    if (!(GEMM_ABL & 2)) { if constexpr (F) swrite_fast(RW, stage ^ 1); else swrite(RW, stage ^ 1); }
    mma(F0);
#pragma unroll
    for (int i = 0; i < NFR; i++) {
      GEMM_SGB(SG_MFMA, 1, 0);
      GEMM_SGB(SG_DS_READ, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      GEMM_SGB(SG_MFMA, SPREAD, 0);
      if (!F) GEMM_SGB(SG_VALU, 12, 0);
      GEMM_SGB(SG_DS_WRITE, 1, 0);
    }
    GEMM_SGB(SG_MFMA, NMFMA, 0);
    if (!(GEMM_ABL & 4)) __syncthreads();
    if (!(GEMM_ABL & 8)) fread(F0, stage ^ 1, 0);
    if (!(GEMM_ABL & 1)) { if constexpr (F) gload_fast(RW, (GEMM_ABL & 16) ? 0 : k_next3); else gload(RW, (GEMM_ABL & 16) ? 0 : k_next3); }
    mma(F1);
#pragma unroll
    for (int i = 0; i < NFR; i++) {
      GEMM_SGB(SG_MFMA, 1, 0);
      GEMM_SGB(SG_DS_READ, 1, 0);
    }
    // the global loads are pinned HERE: left to itself the scheduler sinks them (and their address arithmetic) to the end of the region,
    // i.e. behind the NEXT step's LDS writes — half a step before their data is needed instead of a step and a half
#pragma unroll
    for (int i = 0; i < NPASS; i++) {
      GEMM_SGB(SG_MFMA, SPREAD, 0);
      GEMM_SGB(SG_VALU, F ? 2 : GEMM_LOAD_VALU, 0);
      GEMM_SGB(SG_VMEM_READ, 1, 0);
    }
    GEMM_SGB(SG_MFMA, NMFMA, 0);
  };
  const int nk = (K + GEMM_BK - 1) / GEMM_BK;
#ifdef GEMM_PROF
  unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
  Stage R0, R1;
  Frag F0, F1;
  gload(R0, 0);
  swrite(R0, 0);
  gload(R1, GEMM_BK);
  gload(R0, 2 * GEMM_BK);
  __syncthreads();
  fread(F0, 0, 0);
#ifdef GEMM_PROF
  unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  // two K steps per trip and the odd last one OUTSIDE the loop: with `if (kt + 1 < nk)` around the second step the waitcnt pass has to merge
  // "second step skipped" into the loop head, where the loads of R1 are then the youngest in flight: it drained every load (vmcnt(0)) before
  // the first step's LDS writes, i.e. waited for loads issued half a step earlier (measured: 102 -> see DESIGN.md TF at 20480 x 512 x 1024)
  // Fast steps while neither the tile a step WRITES (kt + 1, and kt + 2 by the second step of a trip) nor the one it LOADS can be the
  // partial tile nk - 1: all steps when K is a multiple of 32, all but the last four (rounded to whole trips) otherwise.  The masked
  // steps behind them may write a fast-loaded register set: its masks are preset to "all valid".
  constexpr std::true_type FAST_STEP{};
  constexpr std::false_type MASKED_STEP{};
  int kt = 0;
  if (FAST) {
    const int nfast = (K % GEMM_BK) == 0 ? nk : max(0, (nk - 4) & ~1);
    for (; kt + 1 < nfast; kt += 2) {
      kstep(FAST_STEP, F0, F1, R1, 0, (kt + 3) * GEMM_BK);          // tile kt in stage 0; tile kt + 1 (R1) goes to stage 1; tile kt + 3 is loaded into R1
      kstep(FAST_STEP, F0, F1, R0, 1, (kt + 4) * GEMM_BK);
    }
    if (kt < nfast) { kstep(FAST_STEP, F0, F1, R1, 0, (kt + 3) * GEMM_BK); kt = nk; }      // (odd nfast only when nfast == nk: this was the last step)
  }
  for (; kt + 1 < nk; kt += 2) {
    kstep(MASKED_STEP, F0, F1, R1, 0, (kt + 3) * GEMM_BK);
    kstep(MASKED_STEP, F0, F1, R0, 1, (kt + 4) * GEMM_BK);
  }
  if (kt < nk) kstep(MASKED_STEP, F0, F1, R1, 0, (kt + 3) * GEMM_BK);
#ifdef GEMM_PROF
  unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
  constexpr bool LN = EPI == 1;
  if constexpr (LN) {
    // rows of the tile are whole rows of the layer (N == BN).  Lane (li, kq) of wave w holds, of the rows 16 a + li, the columns
    // nw + 16 b + 4 kq + r.  Row sums: over r and b in registers, over kq by two cross-lane adds, over the waves through LDS (the K loop's
    // stages are dead).
    constexpr int NW = GemmCfg<NIW>::NWAVE;
    static_assert(2 * BM * NW <= 2 * STAGE, "reduction scratch must fit the K loop's LDS");
    float *red1 = gemm_lds, *red2 = gemm_lds + BM * NW;
    gf4 bv[NI], gv[NI], bev[NI];
#pragma unroll
    for (int b = 0; b < NI; b++) {
      const int col = nw + 16 * b + 4 * kq;
      bv[b] = *reinterpret_cast<const gf4 *>(bias + col); gv[b] = *reinterpret_cast<const gf4 *>(ln.gamma + col); bev[b] = *reinterpret_cast<const gf4 *>(ln.beta + col);
    }
    // z first (straight from the accumulators), then the accumulators are overwritten by silu(z + bias)
    float stat[MT];
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      float p = 0.f;
#pragma unroll
      for (int b = 0; b < NI; b++) {
        if (row < M) *reinterpret_cast<gf4 *>(C + (long long)row * ldc + nw + 16 * b + 4 * kq) = acc[a][b];
#pragma unroll
        for (int r = 0; r < 4; r++) { const float v = acc[a][b][r] + bv[b][r]; acc[a][b][r] = tm_silu(v); p += acc[a][b][r]; }
      }
      p += __shfl_xor(p, 16);
      stat[a] = p + __shfl_xor(p, 32);
    }
    __syncthreads();                   // every wave has read its last fragments
    auto exchange = [&](float *red) {  // stat[a] <- sum over the waves
      if (kq == 0) {
#pragma unroll
        for (int a = 0; a < MT; a++) red[(16 * a + li) * NW + wave] = stat[a];
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const gf4 *q = reinterpret_cast<const gf4 *>(red + (16 * a + li) * NW);
        float m = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NW / 4; w4++) { const gf4 v = q[w4]; m += (v.x + v.y) + (v.z + v.w); }
        stat[a] = m;
      }
    };
    exchange(red1);
    const float inv_n = 1.f / (float)BN;
    float mean[MT];
#pragma unroll
    for (int a = 0; a < MT; a++) {
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
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      const float rstd = rsqrtf(stat[a] * inv_n + ln.eps);
      if (row < M) {
#pragma unroll
        for (int b = 0; b < NI; b++) *reinterpret_cast<gf4 *>(ln.y + (long long)row * ldc + nw + 16 * b + 4 * kq) = acc[a][b] * rstd * gv[b] + bev[b];
        if (wave == 0 && kq == 0) { ln.stats[2 * (long long)row] = mean[a]; ln.stats[2 * (long long)row + 1] = rstd; }
      }
    }
    return;
  }
  // stored straight from the accumulators (routing the tile through LDS to store whole rows was tried and is 6 % SLOWER for these tiles)
  if constexpr (EPI == 2) {
    static_assert(!BT && NI == 2 && NIW == 4, "LayerNorm-backward epilogue: input-gradient kernel, 256-column tile");
    constexpr int NW = GemmCfg<NIW>::NWAVE;
    static_assert(2 * BM * NW <= 2 * STAGE, "reduction scratch must fit the K loop's LDS");
    float *red = gemm_lds;                                   // [80 rows][NW][2]
    const int c0 = nw + 8 * kq;                              // the lane's eight columns (of rows 16 a + li)
    float bv[8], gv[8];
    {
      const gf4 b0 = *reinterpret_cast<const gf4 *>(bias + c0), b1 = *reinterpret_cast<const gf4 *>(bias + c0 + 4);
      const gf4 g0 = *reinterpret_cast<const gf4 *>(ln.gamma + c0), g1 = *reinterpret_cast<const gf4 *>(ln.gamma + c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; j++) { bv[j] = b0[j]; bv[4 + j] = b1[j]; gv[j] = g0[j]; gv[4 + j] = g1[j]; }
    }
    float mean[MT], rstd[MT], m1[MT], m2[MT];      // (z is loaded twice, once per pass: keeping 40 more values live spilled registers)
#define DY(a, j) acc[a][(j) & 1][(j) >> 1]      /* the tile entry of column c0 + j */
    const float inv_n = 1.f / (float)BN;
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      const bool ok = row < M;
      const long long rr = ok ? row : M - 1;
      const gf4 z0 = *reinterpret_cast<const gf4 *>(ln.z + rr * ldc + c0), z1 = *reinterpret_cast<const gf4 *>(ln.z + rr * ldc + c0 + 4);
      mean[a] = ln.stats[2 * rr]; rstd[a] = ln.stats[2 * rr + 1];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (!ok) DY(a, j) = 0.f;                            // (rows past M contribute nothing to the column sums)
        const float v = (j < 4 ? z0[j] : z1[j - 4]) + bv[j];
        const float sig = tm_sigmoid(v), ah = (v * sig - mean[a]) * rstd[a], da = DY(a, j) * gv[j];
        s1 += da; s2 += da * ah;
      }
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      m1[a] = s1 + __shfl_xor(s1, 32); m2[a] = s2 + __shfl_xor(s2, 32);
    }
    __syncthreads();                   // every wave has read its last fragments
    if (kq == 0) {
#pragma unroll
      for (int a = 0; a < MT; a++) *reinterpret_cast<gf2 *>(red + ((16 * a + li) * NW + wave) * 2) = gf2{m1[a], m2[a]};
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const gf4 *q = reinterpret_cast<const gf4 *>(red + (16 * a + li) * NW * 2);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NW / 2; w4++) { const gf4 x = q[w4]; t1 += x.x + x.z; t2 += x.y + x.w; }
      m1[a] = t1 * inv_n; m2[a] = t2 * inv_n;
    }
    float cg[8], cb[8], cz[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { cg[j] = 0.f; cb[j] = 0.f; cz[j] = 0.f; }
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      const long long rr = row < M ? row : M - 1;
      const gf4 z0 = *reinterpret_cast<const gf4 *>(ln.z + rr * ldc + c0), z1 = *reinterpret_cast<const gf4 *>(ln.z + rr * ldc + c0 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float v = (j < 4 ? z0[j] : z1[j - 4]) + bv[j];
        const float sig = tm_sigmoid(v), ah = (v * sig - mean[a]) * rstd[a], da = DY(a, j) * gv[j];
        const float dact = rstd[a] * (da - m1[a] - ah * m2[a]);
        o[j] = dact * (sig * (1.f + v * (1.f - sig)));
        cg[j] += DY(a, j) * ah; cb[j] += DY(a, j); cz[j] += row < M ? o[j] : 0.f;
      }
      if (row < M) {
        float *dst = C + (long long)row * ldc + c0;
        *reinterpret_cast<gf4 *>(dst) = gf4{o[0], o[1], o[2], o[3]}; *reinterpret_cast<gf4 *>(dst + 4) = gf4{o[4], o[5], o[6], o[7]};
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { cg[j] = gemm_row16_sum(cg[j]); cb[j] = gemm_row16_sum(cb[j]); cz[j] = gemm_row16_sum(cz[j]); }
    if (li == 0) {
      float *pp = ln.partial + (size_t)blockIdx.x * 3 * BN + c0;
      *reinterpret_cast<gf4 *>(pp) = gf4{cg[0], cg[1], cg[2], cg[3]}; *reinterpret_cast<gf4 *>(pp + 4) = gf4{cg[4], cg[5], cg[6], cg[7]};
      *reinterpret_cast<gf4 *>(pp + BN) = gf4{cb[0], cb[1], cb[2], cb[3]}; *reinterpret_cast<gf4 *>(pp + BN + 4) = gf4{cb[4], cb[5], cb[6], cb[7]};
      *reinterpret_cast<gf4 *>(pp + 2 * BN) = gf4{cz[0], cz[1], cz[2], cz[3]}; *reinterpret_cast<gf4 *>(pp + 2 * BN + 4) = gf4{cz[4], cz[5], cz[6], cz[7]};
    }
#undef DY
    return;
  }
  if constexpr (!BT && NI == 2) {
    // interleaved tile columns (fread) + transposed tiles (mma): register r of tile b is C[16 a + li][nw + 8 kq + 2 r + b], i.e. the lane
    // holds columns nw + 8 kq .. + 7 of the rows 16 a + li: two dwordx4 per row
    const int col = n0 + nw + 8 * kq;
    const bool vec = !(ldc & 3) && !((uintptr_t)C & 15) && col + 7 < N, vec2 = !(ldc & 1) && !((uintptr_t)C & 7);
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      if (row < M) {
        float *o = C + (long long)row * ldc + col;
        float v[8] = {acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1], acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]};
        if constexpr (EPI == 4) {       // SiLU backward of the producing layer on the accumulators (see the generic path below): `bias` is that layer's
          const float *zr = ln.z + (long long)row * ldc + col;
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (col + k < N) { const float x = zr[k] + bias[col + k], sig = tm_sigmoid(x); v[k] *= sig * (1.f + x * (1.f - sig)); }
          }
        }
        if (vec) { *reinterpret_cast<gf4 *>(o) = gf4{v[0], v[1], v[2], v[3]}; *reinterpret_cast<gf4 *>(o + 4) = gf4{v[4], v[5], v[6], v[7]}; }
        else if (vec2) {                 // rows 8-byte aligned only (an odd multiple of 2 floats wide, e.g. 470 or 286)
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            if (col + k + 1 < N) *reinterpret_cast<gf2 *>(o + k) = gf2{v[k], v[k + 1]};
            else if (col + k < N) o[k] = v[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) if (col + k < N) o[k] = v[k];
        }
      }
    }
    return;
  }
  if constexpr (BT) {                  // transposed tiles (see mma): one dwordx4 per tile and lane where the row is 16-byte aligned
    const bool vec = !(ldc & 3) && !((uintptr_t)C & 15);
#pragma unroll
    for (int b = 0; b < NI; b++) {
      const int col = n0 + nw + 16 * b + 4 * kq;
      gf4 bv = {0.f, 0.f, 0.f, 0.f};
#ifndef GEMM_PROF
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; r++) bv[r] = col + r < N ? bias[col + r] : 0.f;
      }
#endif
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const int row = m0 + 16 * a + li;
        if (row < M) {
          float *o = C + (long long)row * ldc + col;
          const gf4 v = acc[a][b] + bv;
          if constexpr (EPI == 3) {
            // Dense -> SiLU (brax value MLP, ppo_networks.py:180-184): z = acc WITHOUT the bias to C (what tmjx_silu_bwd expects), y = silu(z + bias) to ln.y
            float *yo = ln.y + (long long)row * ldc + col;
            gf4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = tm_silu(v[r]);
            if (vec && col + 3 < N) { *reinterpret_cast<gf4 *>(o) = acc[a][b]; *reinterpret_cast<gf4 *>(yo) = y; }
            else {
#pragma unroll
              for (int r = 0; r < 4; r++) if (col + r < N) { o[r] = acc[a][b][r]; yo[r] = y[r]; }
            }
          } else
          if (vec && col + 3 < N) *reinterpret_cast<gf4 *>(o) = v;
          else {
#pragma unroll
            for (int r = 0; r < 4; r++) if (col + r < N) o[r] = v[r];
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int b = 0; b < NI; b++) {
    const int col = n0 + nw + NI * li + b;
#ifdef GEMM_PROF
    const float bv = 0.f;
#else
    const float bv = (bias && col < N) ? bias[col] : 0.f;
#endif
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * a + 4 * kq + r;
        if constexpr (EPI == 4) {
          // the tile is d loss / d y of a Dense -> SiLU layer (this GEMM: its consumer's input gradient); `bias` is THAT layer's bias, ln.z its saved
          // pre-activation (dense [M][ldc], without the bias): d loss / d z = dy silu'(z + bias) — the expression of k_silu_bwd_f32, whose launch and
          // whose re-read of dy disappear
          if (row < M && col < N) {
            const float v = ln.z[(long long)row * ldc + col] + bv, sig = tm_sigmoid(v);
            C[(long long)row * ldc + col] = acc[a][b][r] * (sig * (1.f + v * (1.f - sig)));
          }
        } else
        if (row < M && col < N) C[(long long)row * ldc + col] = acc[a][b][r] + bv;
      }
  }
  }
#ifdef GEMM_PROF
  if (lane == 0) {
    unsigned long long ts3 = __builtin_amdgcn_s_memtime();
    unsigned long long *o = (unsigned long long *)bias + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * (NT / 64) + wave) * 4;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
  }
#endif
}

#ifndef GEMM_ACT_ONLY      // (csrc/mlp_chain.h shares this header's helpers from a translation unit of its own: the kernels below live in tmjx_hip.hip only)
// ---- dW = dY^T X (+ db as the column of ones): slabs over row ranges
#define DW_BT 128            // output tile: 128 (n) x 128 (k)
#define DW_BM 32             // rows of M per LDS stage
#define DW_LD (DW_BT + 16)   // floats per LDS row: 16 mod 32, lanes of kq and kq + 1 read banks 16 apart
template <bool YVEC, bool XVEC>
__device__ __forceinline__ void gemm_dw_tile(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx, float *__restrict__ slabs,
                                             int M, int N, int K, int with_bias, int rows_per_split, int ld_slab, int tile_n, int tile_k, int split) {
  // 8 waves (two per SIMD), each 64 (n) x 32 (k) of the 128 x 128 tile: 4 x 2 tiles of 16 x 16.  Same pipeline as k_gemm_act: two register
  // sets for the staged rows, ONE barrier in the middle of a 32-row step, MFMAs on both sides of it with the fragment reads, the
  // masking + LDS writes and the global loads spread between them.
  constexpr int STAGE = 2 * DW_BM * DW_LD;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4;
  const int n0 = tile_n * DW_BT, k0 = tile_k * DW_BT;
  const int r_begin = split * rows_per_split, r_end = min(M, r_begin + rows_per_split);
  const int wn = (wave >> 2) * 64, wk = (wave & 3) * 32;
  struct Stage { gf4 y[2], x[2]; int my[2], mx[2]; };
  gf4 colacc = {0.f, 0.f, 0.f, 0.f};        // sum over this thread's rows of its four dY columns (bias gradient)
  auto gload = [&](Stage &R, int r0) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const int f = t + 512 * p, r = f >> 5, c4 = f & 31;           // 32 rows x 32 float4
      const bool ok = r0 + r < r_end;
      R.y[p] = gemm_ld4<YVEC>(dY, (long long)(ok ? r0 + r : r_begin) * ldy + n0, 4 * c4, N - n0, ok, R.my[p]);
      R.x[p] = gemm_ld4<XVEC>(X, (long long)(ok ? r0 + r : r_begin) * ldx + k0, 4 * c4, K - k0, ok, R.mx[p]);
    }
  };
  auto swrite = [&](const Stage &R, int stage) {
    float *sy = gemm_lds + stage * STAGE, *sx = sy + DW_BM * DW_LD;
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const int f = t + 512 * p, r = f >> 5, c4 = f & 31;
      const gf4 vy = gemm_mask4(R.y[p], R.my[p]);
      colacc += vy;
      *reinterpret_cast<gf4 *>(sy + r * DW_LD + 4 * c4) = vy;
      *reinterpret_cast<gf4 *>(sx + r * DW_LD + 4 * c4) = gemm_mask4(R.x[p], R.mx[p]);
    }
  };
  // FAST forms for the 32-row steps that lie entirely inside the slab's row range (see k_gemm_act): no masks; columns outside the matrix are
  // clamped to valid ones (their products land in tile entries that are never stored), only ROWS beyond r_end must be exact zeros (they are
  // summed over) and those exist only in the slab's last, partial step.
  constexpr bool FAST = YVEC && XVEC;
  const float *py[2], *px[2];
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const int f = t + 512 * p, r = f >> 5, c4 = f & 31;
    py[p] = dY + (long long)r * ldy + min(n0 + 4 * c4, ((N + 3) & ~3) - 4);
    px[p] = X + (long long)r * ldx + min(k0 + 4 * c4, ((K + 3) & ~3) - 4);
  }
  auto gload_fast = [&](Stage &R, int r0) {
    const int r0c = r0 + DW_BM <= r_end ? r0 : r_begin;      // steps past the end are never consumed: any valid rows will do
#pragma unroll
    for (int p = 0; p < 2; p++) {
      R.y[p] = *reinterpret_cast<const gf4 *>(py[p] + (long long)r0c * ldy);
      R.x[p] = *reinterpret_cast<const gf4 *>(px[p] + (long long)r0c * ldx);
    }
  };
  auto swrite_fast = [&](const Stage &R, int stage, bool live) {      // live (wave-uniform): the step written lies inside the row range
    float *sy = gemm_lds + stage * STAGE, *sx = sy + DW_BM * DW_LD;
    const float w = live ? 1.f : 0.f;          // (steps past the end hold clamped, i.e. real and finite, rows: they must not reach the bias sum)
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const int f = t + 512 * p, r = f >> 5, c4 = f & 31;
      colacc += w * R.y[p];
      *reinterpret_cast<gf4 *>(sy + r * DW_LD + 4 * c4) = R.y[p];
      *reinterpret_cast<gf4 *>(sx + r * DW_LD + 4 * c4) = R.x[p];
    }
  };
  struct Frag { float a[4][4], b[4][2]; };      // [s-step][tile]
  auto fread = [&](Frag &F, int stage, int half) {
    const float *sy = gemm_lds + stage * STAGE, *sx = sy + DW_BM * DW_LD;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int row = 16 * half + 4 * s + kq;
      // INTERLEAVED tile columns: tile a of the dY side holds columns wn + 4 li + a, tile b of the X side columns wk + 2 li + b, so the
      // four (two) tiles' values of one row are consecutive floats: one ds_read_b128 + one ds_read_b64 per row instead of six ds_read_b32
      const gf4 va = *reinterpret_cast<const gf4 *>(sy + row * DW_LD + wn + 4 * li);
      const gf2 vb = *reinterpret_cast<const gf2 *>(sx + row * DW_LD + wk + 2 * li);
      F.a[s][0] = va.x; F.a[s][1] = va.y; F.a[s][2] = va.z; F.a[s][3] = va.w;
      F.b[s][0] = vb.x; F.b[s][1] = vb.y;
    }
  };
  gf4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const Frag &F) {
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[s][a], F.b[s][b], acc[a][b], 0, 0, 0);
  };
  // The same pinned interleave as k_gemm_act (32 MFMAs per half step and wave): the fragment reads of the other half one per MFMA, then the
  // LDS writes (phase A) / global loads (phase B) spread over the next MFMAs.  Left to itself the compiler clusters the reads in front of
  // their first use (lgkmcnt waits between the MFMA groups) and drains the writes right in front of the barrier.
#ifndef DW_SGB
#define DW_SGB 1
#endif
  auto mstep = [&](auto fast_tag, Frag &F0, Frag &F1, Stage &RW, int stage, int r_next3) {
    constexpr bool F = decltype(fast_tag)::value;
    fread(F1, stage, 1);
    if constexpr (F) swrite_fast(RW, stage ^ 1, r_next3 - 2 * DW_BM < r_end); else swrite(RW, stage ^ 1);
    mma(F0);
    if (DW_SGB) {
#pragma unroll
      for (int i = 0; i < 8; i++) { GEMM_SGB(SG_MFMA, 1, 0); GEMM_SGB(SG_DS_READ, 1, 0); }
#pragma unroll
      for (int i = 0; i < 4; i++) { GEMM_SGB(SG_MFMA, 4, 0); GEMM_SGB(SG_VALU, F ? 4 : 12, 0); GEMM_SGB(SG_DS_WRITE, 1, 0); }
      GEMM_SGB(SG_MFMA, 32, 0);
    }
    __syncthreads();
    fread(F0, stage ^ 1, 0);
    if constexpr (F) gload_fast(RW, r_next3); else gload(RW, r_next3);
    mma(F1);
    if (DW_SGB) {
#pragma unroll
      for (int i = 0; i < 8; i++) { GEMM_SGB(SG_MFMA, 1, 0); GEMM_SGB(SG_DS_READ, 1, 0); }
#pragma unroll
      for (int i = 0; i < 4; i++) { GEMM_SGB(SG_MFMA, 4, 0); GEMM_SGB(SG_VALU, F ? 2 : GEMM_LOAD_VALU, 0); GEMM_SGB(SG_VMEM_READ, 1, 0); }
      GEMM_SGB(SG_MFMA, 32, 0);
    }
  };
  const int nt = (r_end - r_begin + DW_BM - 1) / DW_BM;
  if (nt > 0) {
    Stage R0, R1;
    Frag F0, F1;
    gload(R0, r_begin);
    swrite(R0, 0);
    gload(R1, r_begin + DW_BM);
    gload(R0, r_begin + 2 * DW_BM);
    __syncthreads();
    fread(F0, 0, 0);
    // (two steps per trip, the odd last one outside the loop, fast steps first: see k_gemm_act)
    constexpr std::true_type FAST_STEP{};
    constexpr std::false_type MASKED_STEP{};
    int it = 0;
    if (FAST) {
      const int nfast = ((r_end - r_begin) % DW_BM) == 0 ? nt : max(0, (nt - 4) & ~1);
      for (; it + 1 < nfast; it += 2) {
        mstep(FAST_STEP, F0, F1, R1, 0, r_begin + (it + 3) * DW_BM);
        mstep(FAST_STEP, F0, F1, R0, 1, r_begin + (it + 4) * DW_BM);
      }
      if (it < nfast) { mstep(FAST_STEP, F0, F1, R1, 0, r_begin + (it + 3) * DW_BM); it = nt; }
    }
    for (; it + 1 < nt; it += 2) {
      mstep(MASKED_STEP, F0, F1, R1, 0, r_begin + (it + 3) * DW_BM);
      mstep(MASKED_STEP, F0, F1, R0, 1, r_begin + (it + 4) * DW_BM);
    }
    if (it < nt) mstep(MASKED_STEP, F0, F1, R1, 0, r_begin + (it + 3) * DW_BM);
  }
  float *out = slabs + (size_t)split * (size_t)N * ld_slab;
  if (with_bias && tile_k == 0) {          // (uniform per workgroup) thread t holds columns 4 (t & 31) .. + 3 of the rows t / 32 + 16 j
    __syncthreads();                           // every wave is done with the stages
    float *red = gemm_lds;
    *reinterpret_cast<gf4 *>(red + (t >> 5) * DW_BT + 4 * (t & 31)) = colacc;
    __syncthreads();
    if (t < DW_BT) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) v += red[j * DW_BT + t];
      if (n0 + t < N) out[(size_t)(n0 + t) * ld_slab + K] = v;
    }
  }
  // accumulator register r of tile (a, b) holds output row n0 + wn + 4 (4 kq + r) + a, column k0 + wk + 2 li + b (interleaved tiles, see fread)
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = n0 + wn + 4 * (4 * kq + r) + a, col = k0 + wk + 2 * li;
      if (row < N) {
        float *o = out + (size_t)row * ld_slab + col;
        if (col + 1 < K) *reinterpret_cast<gf2 *>(o) = gf2{acc[a][0][r], acc[a][1][r]};        // (slab rows and the tile origin are 8-byte aligned)
        else if (col < K) o[0] = acc[a][0][r];
      }
    }
}
template <bool YVEC, bool XVEC>
__global__ __launch_bounds__(512) void k_gemm_dw(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx, float *__restrict__ slabs,
                                                 int M, int N, int K, int with_bias, int rows_per_split, int ld_slab) {
  gemm_dw_tile<YVEC, XVEC>(dY, ldy, X, ldx, slabs, M, N, K, with_bias, rows_per_split, ld_slab, blockIdx.x, blockIdx.y, blockIdx.z);
}
// dW[n][k] = sum over slabs; db[n] = the slabs' column K.  One lane per output element, slabs walked with 8 loads in flight.
__device__ __forceinline__ void dw_reduce_elem(const float *__restrict__ slabs, float *__restrict__ dW, float *__restrict__ db, int S, int N, int K,
                                               int with_bias, int ld_slab, int lddw, long long i) {
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
__global__ __launch_bounds__(256) void k_dw_reduce(const float *__restrict__ slabs, float *__restrict__ dW, float *__restrict__ db, int S, int N, int K,
                                                   int with_bias, int ld_slab, int lddw) {
  dw_reduce_elem(slabs, dW, db, S, N, K, with_bias, ld_slab, lddw, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ---- all weight gradients of a backward pass as ONE launch (+ one reduction launch): the per-layer problems are independent once the
// backward chain has produced every dY, each is a single round of workgroups with a short K loop (5 .. 20 row tiles), so launched one by
// one their fixed costs (launch gap, pipeline fill, drain, the reduction's own launch) add up to a third of the time; side by side the
// workgroups of one problem fill the gaps of the next.  The problem table travels BY VALUE in the kernel arguments (hipGraph-safe).
#define DW_GROUP_MAX 16
struct DwProblem {
  const float *dY, *X;
  float *dW, *db, *slabs;
  int ldy, ldx, lddw, M, N, K, rows_per_split, S, ld_slab, tiles_n, tiles_k;
  int wg_begin, red_begin;          // first workgroup of this problem in the grouped launch / in the grouped reduction
};
struct DwGroup { DwProblem p[DW_GROUP_MAX]; int n; };
__global__ __launch_bounds__(512) void k_gemm_dw_grouped(const DwGroup G) {
  int i = 0;
  for (int j = 1; j < G.n; j++) if ((int)blockIdx.x >= G.p[j].wg_begin) i = j;       // uniform
  const DwProblem &P = G.p[i];
  const int local = blockIdx.x - P.wg_begin, tiles = P.tiles_n * P.tiles_k, split = local / tiles, tl = local % tiles;
  gemm_dw_tile<true, true>(P.dY, P.ldy, P.X, P.ldx, P.slabs, P.M, P.N, P.K, P.db != nullptr, P.rows_per_split, P.ld_slab, tl / P.tiles_k, tl % P.tiles_k, split);
}
__global__ __launch_bounds__(256) void k_dw_reduce_grouped(const DwGroup G) {
  int i = 0;
  for (int j = 1; j < G.n; j++) if ((int)blockIdx.x >= G.p[j].red_begin) i = j;
  const DwProblem &P = G.p[i];
  dw_reduce_elem(P.slabs, P.dW, P.db, P.S, P.N, P.K, P.db != nullptr, P.ld_slab, P.lddw, (long long)(blockIdx.x - P.red_begin) * 256 + threadIdx.x);
}
#endif  // GEMM_ACT_ONLY
