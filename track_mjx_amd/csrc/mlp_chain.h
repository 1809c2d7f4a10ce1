// csrc/mlp_chain.h — whole-MLP-chain kernels of the PPO learner for the nets whose hidden layers are ONE GEMM tile (256 columns) wide
// (BASELINE configs[1] / configs[2]: encoder / decoder / critic = [256, 256]; reference layers: flax nn.Dense -> silu -> LayerNorm blocks in
// track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76, brax's value MLP ppo_networks.py:180-184).
//
// One workgroup keeps its 80 (or 32) rows on the CU through Dense -> SiLU -> LayerNorm -> Dense -> ...: a layer's output y goes from the
// accumulators to global memory (the backward pass needs z, y and the row statistics exactly as the layer-by-layer kernels of
// gemm_kernels.h leave them) AND to an LDS image, from which the next layer's MFMAs take their A fragments; only the weight K-slabs are
// streamed (global -> registers -> LDS, two stages, the pipeline of k_gemm_act).  The next layer's first weight tiles are requested
// BEFORE the epilogue of the current one, so neither the weight latency nor the z / y stores sit between two K loops; there is no
// kernel boundary, no pipeline fill / drain and no re-read of y between the layers of a chain.
//
// Arithmetic: every output element is the same k-ordered MFMA chain as in k_gemm_act (chunks of 16 k in order; inside a chunk MFMA e takes
// k = e, 4 + e, 8 + e, 12 + e), zero padding for k >= K only in the last partial tile, and the epilogues are the expressions of
// k_gemm_act's EPI 1 / 2 / 3 / 4 in the same order: results are BIT-IDENTICAL to the layer-by-layer path (tests/test_gpu_chain.py).
//
// LDS (floats; MT = 5: 158 720 B, one workgroup per CU):
//   [0, BM * 256)                 Y image: the current layer's input rows, [row][256], 16-byte chunks XOR-swizzled by (row & 15) — the 16
//                                 lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md) hit 16 distinct slots;
//                                 during the FIRST layer of a chain (A from global memory) the region holds the two A stages [88][40]
//   [.., + 2 * 32 * 260)          weight stages: forward [n][32] with chunks swizzled by (n >> 1) & 7; input gradient [k][BN + 4]
//   [.., + 4 * BM * 8)            row-reduction scratch of the epilogues
#pragma once
#define GEMM_ACT_ONLY
#include "gemm_kernels.h"

#define CH_NT 512
#define CH_WSTAGE (32 * 260)
template <int MT> struct ChainLds {
  static constexpr int BM = 16 * MT, YIMG = BM * 256, RED = 4 * BM * 8, TOTAL = YIMG + 2 * CH_WSTAGE + RED;
  static_assert(2 * (BM + 8) * GEMM_LDA <= YIMG, "the first layer's A stages live in the Y image region");
};

// One GEMM of a chain on the workgroup's row tile: acc[MT][NI] (+)= A[BM rows][K] . op(W), 8 waves side by side along N (16 NI columns each).
//   BT     W is [N][K] (forward: y = x W^T), LDS image [n][32] swizzled;  !BT W is [K][N] (input gradient: dx = dy W), LDS image [k][BN + 4]
//   A_LDS  the A operand is the Y image (K == 256); otherwise A is read from global memory and staged like k_gemm_act's
template <int MT, int NI, bool BT, bool A_LDS>
struct ChainGemm {
  static constexpr int BM = 16 * MT, BN = 128 * NI, A_ROWS = BM + 8;
  static constexpr int A_PASS = A_LDS ? 0 : (BM * 8 + CH_NT - 1) / CH_NT, B_PASS = BN / 64, NPASS = A_PASS + B_PASS;
  static constexpr int LDB_N = BN + 4, A_STAGE = A_ROWS * GEMM_LDA;
  static constexpr int NMFMA = 4 * MT * NI, NFR = MT + (BT ? NI : 4 * NI);
  static constexpr int SPREAD = (NMFMA - NFR) / (NPASS + 1) > 0 ? (NMFMA - NFR) / (NPASS + 1) : 1;
  struct Stage { gf4 v[NPASS]; int m[NPASS]; };
  struct Frag { gf4 a[MT], b[NI]; };
  // (A_LDS) the Y image this GEMM reads is ALSO what has to reach global memory (a hidden layer's y, a stage's d loss / d z): the producing epilogue
  // leaves that store to THIS K loop — T4 = BM / 8 float4 per thread, TRK of them per K step, read back from the image (a wave per row: 1 KB
  // contiguous) — so that the 80 KB do not leave the CU as one burst with the matrix pipe idle, but under the MFMAs of the next GEMM
  static constexpr int T4 = A_LDS ? BM / 8 : 0, TRK = MT >= 4 ? 2 : 1;
  const float *A, *W;
  float *ast, *yimg, *wst, *tr_dst;
  int lda, ldw, M, N, K, m0, t, li, kq, nw;
  const float *pa[A_PASS > 0 ? A_PASS : 1], *pb[B_PASS];
  Stage R0, R1;
  Frag F0, F1;

  __device__ __forceinline__ void init(const float *A_, int lda_, int M_, int m0_, const float *W_, int ldw_, int N_, int K_, float *yimg_, float *wst_, float *tr_ = nullptr) {
    A = A_; lda = lda_; M = M_; m0 = m0_; W = W_; ldw = ldw_; N = N_; K = K_; yimg = yimg_; ast = yimg_; wst = wst_;
    tr_dst = tr_ ? tr_ + (long long)m0_ * 256 : nullptr;
    t = threadIdx.x; li = t & 15; kq = (t & 63) >> 4; nw = (t >> 6) * 16 * NI;
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + CH_NT * p, r = f < BM * 8 ? f >> 3 : 0, c4 = f & 7;
      pa[p] = A + (long long)min(m0 + r, M - 1) * lda + 4 * c4;
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++) {
      const int f = t + CH_NT * p;
      if (BT) { const int r = f >> 3, c4 = f & 7; pb[p] = W + (long long)min(r, N - 1) * ldw + 4 * c4; }
      else { const int r = f / (BN / 4), c4 = f % (BN / 4); pb[p] = W + (long long)r * ldw + min(4 * c4, ((N + 3) & ~3) - 4); }
    }
  }
  // masked loads: exact zeros for k >= K (and for rows / columns outside the matrices), as k_gemm_act's gload
  __device__ __forceinline__ void gload(Stage &R, int k0) {
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + CH_NT * p, r = f >> 3, c4 = f & 7;
      const bool ok = f < BM * 8 && m0 + r < M;
      R.v[p] = gemm_ld4<true>(A, (long long)(m0 + (ok ? r : 0)) * lda, k0 + 4 * c4, K, ok, R.m[p]);
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++) {
      const int f = t + CH_NT * p;
      if (BT) {
        const int r = f >> 3, c4 = f & 7;
        const bool ok = r < N;
        R.v[A_PASS + p] = gemm_ld4<true>(W, (long long)(ok ? r : 0) * ldw, k0 + 4 * c4, K, ok, R.m[A_PASS + p]);
      } else {
        const int r = f / (BN / 4), c4 = f % (BN / 4);
        const bool ok = k0 + r < K;
        R.v[A_PASS + p] = gemm_ld4<true>(W, (long long)(ok ? k0 + r : 0) * ldw, 4 * c4, N, ok, R.m[A_PASS + p]);
      }
    }
  }
  __device__ __forceinline__ float *b_slot(float *sb, int f) const {
    if (BT) { const int r = f >> 3, c4 = f & 7; return sb + r * 32 + ((c4 ^ ((r >> 1) & 7)) << 2); }
    const int r = f / (BN / 4), c4 = f % (BN / 4);
    return sb + r * LDB_N + 4 * c4;
  }
  template <bool MASKED>
  __device__ __forceinline__ void swrite(const Stage &R, int stage) {
    float *sa = ast + stage * A_STAGE, *sb = wst + stage * CH_WSTAGE;
#pragma unroll
    for (int p = 0; p < A_PASS; p++) {
      const int f = t + CH_NT * p, r = f < BM * 8 ? f >> 3 : BM + ((f >> 3) & 7), c4 = f & 7;
      *reinterpret_cast<gf4 *>(sa + r * GEMM_LDA + 4 * c4) = MASKED ? gemm_mask4(R.v[p], R.m[p]) : R.v[p];
    }
#pragma unroll
    for (int p = 0; p < B_PASS; p++)
      *reinterpret_cast<gf4 *>(b_slot(sb, t + CH_NT * p)) = MASKED ? gemm_mask4(R.v[A_PASS + p], R.m[A_PASS + p]) : R.v[A_PASS + p];
  }
  __device__ __forceinline__ void gload_fast(Stage &R, int k0) {
    const int k0c = k0 + GEMM_BK <= K ? k0 : 0;        // tiles past the end are never consumed: any valid address will do
#pragma unroll
    for (int p = 0; p < A_PASS; p++) R.v[p] = *reinterpret_cast<const gf4 *>(pa[p] + k0c);
#pragma unroll
    for (int p = 0; p < B_PASS; p++) R.v[A_PASS + p] = *reinterpret_cast<const gf4 *>(pb[p] + (BT ? (long long)k0c : (long long)k0c * ldw));
  }
  // fragments of the 16-deep chunk c of K tile kt (in stage `stage`)
  __device__ __forceinline__ void fread(Frag &F, int stage, int c, int kt) {
    const float *sb = wst + stage * CH_WSTAGE;
    if (A_LDS) {
#pragma unroll
      for (int a = 0; a < MT; a++) F.a[a] = *reinterpret_cast<const gf4 *>(yimg + (16 * a + li) * 256 + (((8 * kt + 4 * c + kq) ^ li) << 2));
    } else {
      const float *sa = ast + stage * A_STAGE;
#pragma unroll
      for (int a = 0; a < MT; a++) F.a[a] = *reinterpret_cast<const gf4 *>(sa + (16 * a + li) * GEMM_LDA + 16 * c + 4 * kq);
    }
    if (BT) {
#pragma unroll
      for (int b = 0; b < NI; b++) F.b[b] = *reinterpret_cast<const gf4 *>(sb + (nw + 16 * b + li) * 32 + (((4 * c + kq) ^ (li >> 1)) << 2));
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) {        // interleaved tile columns (tile b holds columns nw + NI li + b): one ds_read_b64 per k
        const float *q = sb + (16 * c + 4 * kq + e) * LDB_N + nw + NI * li;
        if constexpr (NI == 2) { const gf2 v = *reinterpret_cast<const gf2 *>(q); F.b[0][e] = v.x; F.b[1][e] = v.y; }
        else F.b[0][e] = q[0];
      }
    }
  }
  __device__ __forceinline__ void mma(const Frag &F, gf4 (&acc)[MT][NI]) {
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < NI; b++) {
          // transposed tiles (the weight fragment as the FIRST operand): BT: register r = C[16 a + li][nw + 16 b + 4 kq + r];
          // !BT with two interleaved tiles: register r of tile b = C[16 a + li][nw + 8 kq + 2 r + b];  !BT, NI == 1: native layout
          if constexpr (BT || NI == 2) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.b[b][e], F.a[a][e], acc[a][b], 0, 0, 0);
          else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(F.a[a][e], F.b[b][e], acc[a][b], 0, 0, 0);
        }
  }
  // one K step of tile kt in stage s: k_gemm_act's kstep (one barrier in the middle, MFMAs on both sides, pinned interleave)
  template <bool FASTS, int TR = 0>
  __device__ __forceinline__ void kstep(gf4 (&acc)[MT][NI], Stage &RW, int stage, int kt) {
    gf4 trv[TR > 0 ? TR : 1];
    fread(F1, stage, 1, kt);
    swrite<!FASTS>(RW, stage ^ 1);
    mma(F0, acc);
#pragma unroll
    for (int i = 0; i < NFR; i++) { GEMM_SGB(SG_MFMA, 1, 0); GEMM_SGB(SG_DS_READ, 1, 0); }
#pragma unroll
    for (int i = 0; i < NPASS; i++) { GEMM_SGB(SG_MFMA, SPREAD, 0); if (!FASTS) GEMM_SGB(SG_VALU, 12, 0); GEMM_SGB(SG_DS_WRITE, 1, 0); }
    GEMM_SGB(SG_MFMA, NMFMA, 0);
    __syncthreads();
    fread(F0, stage ^ 1, 0, kt + 1);
    if constexpr (TR > 0) {            // this step's share of the image -> registers (float4 f of the tile: row f / 64 — the wave's own —, chunk f % 64)
#pragma unroll
      for (int j = 0; j < TR; j++) {
        const int f = t + CH_NT * min(kt * TRK + j, T4 - 1), row = f >> 6, c4 = f & 63;
        trv[j] = *reinterpret_cast<const gf4 *>(yimg + row * 256 + ((c4 ^ (row & 15)) << 2));
      }
    }
    if (FASTS) gload_fast(RW, (kt + 3) * GEMM_BK); else gload(RW, (kt + 3) * GEMM_BK);
    mma(F1, acc);
    if constexpr (TR > 0) {            // ... and on to global memory (rows past M land in the buffer's padding rows: no predicate, no branch in the step)
#pragma unroll
      for (int j = 0; j < TR; j++) {
        const int f = t + CH_NT * min(kt * TRK + j, T4 - 1);
        *reinterpret_cast<gf4 *>(tr_dst + (long long)(f >> 6) * 256 + 4 * (f & 63)) = trv[j];
      }
    }
#pragma unroll
    for (int i = 0; i < NFR + TR; i++) { GEMM_SGB(SG_MFMA, 1, 0); GEMM_SGB(SG_DS_READ, 1, 0); }
#pragma unroll
    for (int i = 0; i < NPASS; i++) { GEMM_SGB(SG_MFMA, SPREAD, 0); GEMM_SGB(SG_VALU, FASTS ? 2 : GEMM_LOAD_VALU, 0); GEMM_SGB(SG_VMEM_READ, 1, 0); }
#pragma unroll
    for (int i = 0; i < TR; i++) { GEMM_SGB(SG_MFMA, 1, 0); GEMM_SGB(SG_VALU, 4, 0); GEMM_SGB(SG_VMEM_WRITE, 1, 0); }
    GEMM_SGB(SG_MFMA, NMFMA, 0);
  }
  // The first two weight (and A) tiles are requested EARLY — in a chain: before the previous layer's epilogue, so that their latency and that
  // epilogue's stores never sit in front of this K loop (the loads are older than the stores: waiting for them does not wait for the stores)
  // (a GEMM whose A operand is the Y image has K = 256: whole tiles only — the unmasked forms throughout, no mask registers; rows / columns of W
  // outside the matrix are clamped, they only reach accumulators that are never stored)
  __device__ __forceinline__ void early() {
    if (A_LDS) { gload_fast(R0, 0); gload_fast(R1, GEMM_BK); } else { gload(R0, 0); gload(R1, GEMM_BK); }
  }
  // ... and written once the previous K loop's stages are dead (a barrier separates them from the caller's last fragment reads)
  __device__ __forceinline__ void late() {
    if (A_LDS) { swrite<false>(R0, 0); gload_fast(R0, 2 * GEMM_BK); } else { swrite<true>(R0, 0); gload(R0, 2 * GEMM_BK); }
    __syncthreads();
    fread(F0, 0, 0, 0);
  }
  __device__ __forceinline__ void loop(gf4 (&acc)[MT][NI]) {
    if constexpr (A_LDS) {
      // K = 256: eight unmasked steps; the first NTRS of them (an even number: the steps come in pairs, one per register set) also carry the image
      // to global memory, TRK float4 per thread and step (slots past T4 repeat the last one: the same bytes to the same address)
      constexpr int NTRS = 2 * (((T4 + TRK - 1) / TRK + 1) / 2);
      int kt = 0;
      for (; kt < NTRS; kt += 2) { kstep<true, TRK>(acc, R1, 0, kt); kstep<true, TRK>(acc, R0, 1, kt + 1); }
      for (; kt < 8; kt += 2) { kstep<true, 0>(acc, R1, 0, kt); kstep<true, 0>(acc, R0, 1, kt + 1); }
      return;
    }
    const int nk = (K + GEMM_BK - 1) / GEMM_BK;
    int kt = 0;
    const int nfast = (A_LDS || (K % GEMM_BK) == 0) ? nk : max(0, (nk - 4) & ~1);
    for (; kt + 1 < nfast; kt += 2) { kstep<true>(acc, R1, 0, kt); kstep<true>(acc, R0, 1, kt + 1); }
    if (kt < nfast) { kstep<true>(acc, R1, 0, kt); kt = nk; }
    if constexpr (!A_LDS) {
      for (; kt + 1 < nk; kt += 2) { kstep<false>(acc, R1, 0, kt); kstep<false>(acc, R0, 1, kt + 1); }
      if (kt < nk) kstep<false>(acc, R1, 0, kt);
    }
  }
};

// ---- forward chain
#define CHAIN_MAX_HIDDEN 4
struct ChainHidden { const float *W, *bias, *gamma, *beta; float *z, *y, *stats; int K, ldw; };      // a 256-wide hidden layer; z, y dense [M][256]
struct ChainFwd {
  const float *A; int lda, M, nh;
  ChainHidden h[CHAIN_MAX_HIDDEN];
  const float *Wf, *bf; float *outf; int Nf, ldwf, ldof;      // FIN 1: un-activated last layer (Nf <= 128); FIN 2: the 1-wide head (Wf = its weight row)
  float eps;
  // LAT (FIN 1, the encoder): behind fc2 = [mean | logvar] the latent sample z = mean + lat_eps * exp(logvar / 2) (reparameterize,
  // intention_network.py:84-88) and the decoder's input [z | proprioception] (:128-139) — lat_out[M][lat_ld] <- [z (lat_Z) | prop[M][prop_w] (row stride prop_ld)]
  const float *lat_eps; float *lat_out; const float *prop; int lat_Z, lat_ld, prop_w, prop_ld;
  unsigned long long *prof;      // NULL, or 16 shader-clock stamps per workgroup (tools/chain_stamps.py): kernel start, then behind every K loop and every epilogue
};
#define CHAIN_STAMP(i) do { if (P.prof && threadIdx.x == 0) P.prof[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// y -> the Y image (the next layer's A operand): lane (li, kq) of wave w holds rows 16 a + li, columns nw + 16 b + 4 kq .. + 3 = chunk (nw + 16 b) / 4 + kq
template <int MT>
__device__ __forceinline__ void chain_y_to_lds(float *yimg, const gf4 (&y)[MT][2], int li, int kq, int nw) {
#pragma unroll
  for (int a = 0; a < MT; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) *reinterpret_cast<gf4 *>(yimg + (16 * a + li) * 256 + (((((nw + 16 * b) >> 2) + kq) ^ li) << 2)) = y[a][b];
}

// EPI 1: hidden layers are Dense -> SiLU -> LayerNorm blocks (k_gemm_act<.., EPI = 1>'s epilogue);  EPI 3: Dense -> SiLU (brax value MLP, EPI = 3)
// FIN 0: the chain ends with its last hidden layer;  1: + an un-activated layer of at most 128 columns (fc2 / the policy head);
//     2: + the value MLP's 1-wide head as a dot product, summed in k_head_fwd's order (lane groups of four columns by fmaf, then its xor butterfly)
template <int MT, int EPI, int FIN, bool LAT = false>
__global__ __launch_bounds__(CH_NT) void k_chain_fwd(const ChainFwd P) {
  using LD = ChainLds<MT>;
  constexpr int BM = LD::BM, NW = 8;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];
  float *yimg = gemm_lds, *wst = gemm_lds + LD::YIMG, *red1 = wst + 2 * CH_WSTAGE, *red2 = red1 + BM * NW;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4, nw = wave * 32;
  const int m0 = blockIdx.x * BM, M = P.M;
  gf4 acc[MT][2];
  auto zero = [&]() {
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
  };
  zero();
  int stamp = 0;
  CHAIN_STAMP(stamp++);
  {
    ChainGemm<MT, 2, true, false> G;
    G.init(P.A, P.lda, M, m0, P.h[0].W, P.h[0].ldw, 256, P.h[0].K, yimg, wst);
    G.early(); G.late(); G.loop(acc);
  }
  CHAIN_STAMP(stamp++);
  ChainGemm<MT, 2, true, true> G;
  ChainGemm<MT, 1, true, true> Gf;
  for (int l = 0;; l++) {
    const ChainHidden &H = P.h[l];
    const bool more = l + 1 < P.nh;         // (uniform)
    if (more) { G.init(nullptr, 0, M, m0, P.h[l + 1].W, P.h[l + 1].ldw, 256, 256, yimg, wst, H.y); G.early(); }
    else if (FIN == 1) { Gf.init(nullptr, 0, M, m0, P.Wf, P.ldwf, P.Nf, 256, yimg, wst, H.y); Gf.early(); }
    // ---- epilogue of hidden layer l (the expressions of k_gemm_act's EPI 1 / EPI 3, in their order)
    gf4 bv[2], gv[2], bev[2];
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const int col = nw + 16 * b + 4 * kq;
      bv[b] = *reinterpret_cast<const gf4 *>(H.bias + col);
      if (EPI == 1) { gv[b] = *reinterpret_cast<const gf4 *>(H.gamma + col); bev[b] = *reinterpret_cast<const gf4 *>(H.beta + col); }
    }
    float stat[MT];
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      float p = 0.f;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        if (row < M) *reinterpret_cast<gf4 *>(H.z + (long long)row * 256 + nw + 16 * b + 4 * kq) = acc[a][b];
#pragma unroll
        for (int r = 0; r < 4; r++) { const float v = acc[a][b][r] + bv[b][r]; acc[a][b][r] = tm_silu(v); p += acc[a][b][r]; }
      }
      if (EPI == 1) { p += __shfl_xor(p, 16); stat[a] = p + __shfl_xor(p, 32); }
    }
    __syncthreads();                   // every wave has read its last fragments (Y image and stages are dead)
    if (EPI == 1) {
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
      const float inv_n = 1.f / 256.f;
      float mean[MT];
#pragma unroll
      for (int a = 0; a < MT; a++) {
        mean[a] = stat[a] * inv_n;
        float q = 0.f;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
          for (int r = 0; r < 4; r++) { acc[a][b][r] -= mean[a]; q += acc[a][b][r] * acc[a][b][r]; }
        q += __shfl_xor(q, 16);
        stat[a] = q + __shfl_xor(q, 32);
      }
      exchange(red2);
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const int row = m0 + 16 * a + li;
        const float rstd = rsqrtf(stat[a] * inv_n + P.eps);
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = acc[a][b] * rstd * gv[b] + bev[b];
        if (row < M && wave == 0 && kq == 0) { H.stats[2 * (long long)row] = mean[a]; H.stats[2 * (long long)row + 1] = rstd; }
      }
    }
    // y: to the Y image (the next layer's A operand) — the next GEMM's K loop also carries it to global memory (what the backward pass and the weight
    // gradients read), ChainGemm::kstep — or, behind the chain's last GEMM, to global memory from here
    if (more || FIN == 1) chain_y_to_lds<MT>(yimg, acc, li, kq, nw);
    else {
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const int row = m0 + 16 * a + li;
        if (row < M) {
#pragma unroll
          for (int b = 0; b < 2; b++) *reinterpret_cast<gf4 *>(H.y + (long long)row * 256 + nw + 16 * b + 4 * kq) = acc[a][b];
        }
      }
    }
    CHAIN_STAMP(stamp++);
    if (!more) break;
    zero();
    G.late(); G.loop(acc);
    CHAIN_STAMP(stamp++);
  }
  if (FIN == 1) {
    gf4 accf[MT][1];
#pragma unroll
    for (int a = 0; a < MT; a++) accf[a][0] = gf4{0.f, 0.f, 0.f, 0.f};
    Gf.late(); Gf.loop(accf);
    CHAIN_STAMP(stamp++);
    const int col = wave * 16 + 4 * kq;       // NI == 1: 16 columns per wave
    gf4 bvf = {0.f, 0.f, 0.f, 0.f};
    if (P.bf) {
#pragma unroll
      for (int r = 0; r < 4; r++) bvf[r] = col + r < P.Nf ? P.bf[col + r] : 0.f;
    }
    const bool vec = !(P.ldof & 3) && !((uintptr_t)P.outf & 15);
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      const gf4 v = accf[a][0] + bvf;
      if (row < M) {
        float *o = P.outf + (long long)row * P.ldof + col;
        if (vec && col + 3 < P.Nf) *reinterpret_cast<gf4 *>(o) = v;
        else {
#pragma unroll
          for (int r = 0; r < 4; r++) if (col + r < P.Nf) o[r] = v[r];
        }
      }
      if (LAT) accf[a][0] = v;
    }
    if (LAT) {
      // the fc2 tile through LDS (mean column j and logvar column Z + j sit in different waves), then the decoder input of the tile's rows:
      // k_latent_concat's expression, fmaf(eps, expf(logvar / 2), mean), and the proprioceptive columns copied next to it
      float *ft = yimg;                        // [BM][128]
      __syncthreads();
#pragma unroll
      for (int a = 0; a < MT; a++) *reinterpret_cast<gf4 *>(ft + (16 * a + li) * 128 + col) = accf[a][0];
      __syncthreads();
      const int Z = P.lat_Z, zq = Z >> 2;
      for (int it = t; it < BM * zq; it += CH_NT) {
        const int row = it / zq, q = it - row * zq;
        const long long gr = m0 + row;
        if (gr < M) {
          const gf4 mu = *reinterpret_cast<const gf4 *>(ft + row * 128 + 4 * q), lv = *reinterpret_cast<const gf4 *>(ft + row * 128 + Z + 4 * q);
          const gf4 ep = *reinterpret_cast<const gf4 *>(P.lat_eps + gr * Z + 4 * q);
          gf4 z;
#pragma unroll
          for (int r = 0; r < 4; r++) z[r] = fmaf(ep[r], expf(0.5f * lv[r]), mu[r]);
          *reinterpret_cast<gf4 *>(P.lat_out + gr * P.lat_ld + 4 * q) = z;
        }
      }
      const int ph = P.prop_w >> 1;
      for (int it = t; it < BM * ph; it += CH_NT) {
        const int row = it / ph, q = it - row * ph;
        const long long gr = m0 + row;
        if (gr < M) *reinterpret_cast<gf2 *>(P.lat_out + gr * P.lat_ld + Z + 2 * q) = *reinterpret_cast<const gf2 *>(P.prop + gr * P.prop_ld + 2 * q);
      }
    }
  }
  if (FIN == 2) {
    // out[row] = y[row][:] . w + b in k_head_fwd's order: group g of four consecutive columns = one fmaf chain from 0 (its lane g), then the
    // xor butterfly 32, 16, 8, 4, 2, 1 over the 64 groups — here group g = 8 wave + 4 b + kq sits in lane (li = row, kq) of wave `wave`
    float *hs = wst;                         // [BM][64] group sums (the stages are dead)
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const gf4 wv = *reinterpret_cast<const gf4 *>(P.Wf + nw + 16 * b + 4 * kq);
#pragma unroll
      for (int a = 0; a < MT; a++) {
        float s = 0.f;
        s = fmaf(acc[a][b][0], wv[0], s); s = fmaf(acc[a][b][1], wv[1], s); s = fmaf(acc[a][b][2], wv[2], s); s = fmaf(acc[a][b][3], wv[3], s);
        hs[(16 * a + li) * 64 + 8 * wave + 4 * b + kq] = s;
      }
    }
    __syncthreads();
    if (t < BM && m0 + t < M) {
      float v[64];
#pragma unroll
      for (int g = 0; g < 64; g += 4) { const gf4 x = *reinterpret_cast<const gf4 *>(hs + t * 64 + g); v[g] = x.x; v[g + 1] = x.y; v[g + 2] = x.z; v[g + 3] = x.w; }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int g = 0; g < off; g++) v[g] = v[g] + v[g + off];         // lane g's value after the xor-`off` step (both partners hold the same sum)
      P.outf[m0 + t] = v[0] + (P.bf ? P.bf[0] : 0.f);
    }
  }
  CHAIN_STAMP(stamp++);
}

// ---- backward chain: the input-gradient GEMMs of a chain, each with the PRODUCING block's backward in its epilogue (k_gemm_act's EPI 2 / EPI 4),
// d loss / d z of every hidden layer to global memory (the weight gradients' operand) and to the Y image (the next GEMM's A operand)
struct ChainBwdStage {
  const float *W; int ldw;                       // stage 0: the last layer's weight [Kg][256] (HEAD: its weight row); stage s > 0: hidden layer (nh - s)'s weight [256][256]
  const float *z, *bias, *gamma, *stats;         // the block whose backward this stage's epilogue applies: hidden layer (nh - 1 - s)
  float *dz, *partial;                           // its d loss / d z [M][256]; EPI 2: this workgroup's column sums (d gamma | d beta | d bias) [3][256] at partial + 768 blockIdx.x
};
struct ChainBwd {
  const float *G; int ldg, Kg, M, ns;            // d loss / d (last layer's output) [M][Kg] (Kg <= 128); HEAD: dy1 [M]
  ChainBwdStage s[CHAIN_MAX_HIDDEN];
  const float *W0; int ldw0, dx_cols; float *dx; int lddx;      // DX: + the first hidden layer's input gradient, its first dx_cols (<= 128) columns only
  unsigned long long *prof;
};

// o[0..7] = columns c0 .. c0 + 7 of rows 16 a + li -> the Y image (two 16-byte chunks)
__device__ __forceinline__ void chain_o_to_lds(float *yimg, const float (&o)[8], int a, int li, int c0) {
  float *row = yimg + (16 * a + li) * 256;
  *reinterpret_cast<gf4 *>(row + ((((c0 >> 2)) ^ li) << 2)) = gf4{o[0], o[1], o[2], o[3]};
  *reinterpret_cast<gf4 *>(row + ((((c0 >> 2) + 1) ^ li) << 2)) = gf4{o[4], o[5], o[6], o[7]};
}

// EPI 2: the hidden layers are Dense -> SiLU -> LayerNorm blocks;  EPI 4: Dense -> SiLU layers.
// HEAD: the chain's last layer is the value MLP's 1-wide head — stage 0 has no GEMM, its d loss / d y is the outer product dy1 x (the head's weight row)
// (k_silu_bwd_rank1_f32's expression).  DX: a trailing GEMM for the first hidden layer's input gradient (the decoder: d loss / d latent).
template <int MT, int EPI, bool HEAD, bool DX>
__global__ __launch_bounds__(CH_NT) void k_chain_bwd(const ChainBwd P) {
  using LD = ChainLds<MT>;
  constexpr int BM = LD::BM, NW = 8;
  extern __shared__ __attribute__((aligned(16))) float gemm_lds[];
  float *yimg = gemm_lds, *wst = gemm_lds + LD::YIMG, *red = wst + 2 * CH_WSTAGE;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, kq = lane >> 4, nw = wave * 32;
  const int m0 = blockIdx.x * BM, M = P.M;
  const int c0 = nw + 8 * kq;                              // the lane's eight columns (of rows 16 a + li)
  gf4 acc[MT][2];
#define DY(a, j) acc[a][(j) & 1][(j) >> 1]      /* the tile entry of column c0 + j */
  auto zero = [&]() {
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
  };
  int stamp = 0;
  CHAIN_STAMP(stamp++);
  if (HEAD) {
    float wv[8];
    { const gf4 w0 = *reinterpret_cast<const gf4 *>(P.s[0].W + c0), w1 = *reinterpret_cast<const gf4 *>(P.s[0].W + c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; j++) { wv[j] = w0[j]; wv[4 + j] = w1[j]; } }
#pragma unroll
    for (int a = 0; a < MT; a++) {
      const int row = m0 + 16 * a + li;
      const float d = P.G[row < M ? row : M - 1];
#pragma unroll
      for (int j = 0; j < 8; j++) DY(a, j) = d * wv[j];
    }
  } else {
    zero();
    ChainGemm<MT, 2, false, false> G;
    G.init(P.G, P.ldg, M, m0, P.s[0].W, P.s[0].ldw, 256, P.Kg, yimg, wst);
    G.early(); G.late(); G.loop(acc);
  }
  CHAIN_STAMP(stamp++);
  ChainGemm<MT, 2, false, true> G;
  ChainGemm<MT, 1, false, true> Gx;
  for (int st = 0;; st++) {
    const ChainBwdStage &S = P.s[st];
    const bool more = st + 1 < P.ns;        // (uniform)
    // the next GEMM's first weight tiles are requested in front of the epilogue — except where the epilogue has no registers to carry them through
    // (the 80-row tile's LayerNorm backward: 79 spilled registers with them), there behind it
    constexpr bool PRE = !(MT == 5 && (EPI == 2 || HEAD));
    if (more) { G.init(nullptr, 0, M, m0, P.s[st + 1].W, P.s[st + 1].ldw, 256, 256, yimg, wst, S.dz); if (PRE) G.early(); }
    else if (DX) { Gx.init(nullptr, 0, M, m0, P.W0, P.ldw0, P.dx_cols, 256, yimg, wst, S.dz); if (PRE) Gx.early(); }
    const bool keep = more || DX;           // d loss / d z also feeds a GEMM of this launch
    float bv[8], gv[8];
    {
      const gf4 b0 = *reinterpret_cast<const gf4 *>(S.bias + c0), b1 = *reinterpret_cast<const gf4 *>(S.bias + c0 + 4);
#pragma unroll
      for (int j = 0; j < 4; j++) { bv[j] = b0[j]; bv[4 + j] = b1[j]; }
      if (EPI == 2) {
        const gf4 g0 = *reinterpret_cast<const gf4 *>(S.gamma + c0), g1 = *reinterpret_cast<const gf4 *>(S.gamma + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { gv[j] = g0[j]; gv[4 + j] = g1[j]; }
      }
    }
    if (EPI == 2) {
      // k_gemm_act's EPI 2, expression for expression (z is loaded twice, once per pass)
      float mean[MT], rstd[MT], m1[MT], m2[MT];
      const float inv_n = 1.f / 256.f;
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const int row = m0 + 16 * a + li;
        const bool ok = row < M;
        const long long rr = ok ? row : M - 1;
        const gf4 z0 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0), z1 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0 + 4);
        mean[a] = S.stats[2 * rr]; rstd[a] = S.stats[2 * rr + 1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (!ok) DY(a, j) = 0.f;
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
        const gf4 z0 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0), z1 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0 + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float v = (j < 4 ? z0[j] : z1[j - 4]) + bv[j];
          const float sig = tm_sigmoid(v), ah = (v * sig - mean[a]) * rstd[a], da = DY(a, j) * gv[j];
          const float dact = rstd[a] * (da - m1[a] - ah * m2[a]);
          o[j] = dact * (sig * (1.f + v * (1.f - sig)));
          cg[j] += DY(a, j) * ah; cb[j] += DY(a, j); cz[j] += row < M ? o[j] : 0.f;
        }
        if (keep) chain_o_to_lds(yimg, o, a, li, c0);        // (the consuming GEMM's K loop carries it on to global memory)
        else if (row < M) {
          float *dst = S.dz + (long long)row * 256 + c0;
          *reinterpret_cast<gf4 *>(dst) = gf4{o[0], o[1], o[2], o[3]}; *reinterpret_cast<gf4 *>(dst + 4) = gf4{o[4], o[5], o[6], o[7]};
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) { cg[j] = gemm_row16_sum(cg[j]); cb[j] = gemm_row16_sum(cb[j]); cz[j] = gemm_row16_sum(cz[j]); }
      if (li == 0) {
        float *pp = S.partial + (size_t)blockIdx.x * 3 * 256 + c0;
        *reinterpret_cast<gf4 *>(pp) = gf4{cg[0], cg[1], cg[2], cg[3]}; *reinterpret_cast<gf4 *>(pp + 4) = gf4{cg[4], cg[5], cg[6], cg[7]};
        *reinterpret_cast<gf4 *>(pp + 256) = gf4{cb[0], cb[1], cb[2], cb[3]}; *reinterpret_cast<gf4 *>(pp + 256 + 4) = gf4{cb[4], cb[5], cb[6], cb[7]};
        *reinterpret_cast<gf4 *>(pp + 512) = gf4{cz[0], cz[1], cz[2], cz[3]}; *reinterpret_cast<gf4 *>(pp + 512 + 4) = gf4{cz[4], cz[5], cz[6], cz[7]};
      }
    } else {
      // k_gemm_act's EPI 4 (interleaved-tile path): d loss / d z = dy silu'(z + bias)
      if (keep) __syncthreads();         // every wave has read its last fragments (the Y image is rewritten below)
#pragma unroll
      for (int a = 0; a < MT; a++) {
        const int row = m0 + 16 * a + li;
        const long long rr = row < M ? row : M - 1;
        const gf4 z0 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0), z1 = *reinterpret_cast<const gf4 *>(S.z + rr * 256 + c0 + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float x = (j < 4 ? z0[j] : z1[j - 4]) + bv[j], sig = tm_sigmoid(x);
          o[j] = DY(a, j) * (sig * (1.f + x * (1.f - sig)));
        }
        if (keep) chain_o_to_lds(yimg, o, a, li, c0);        // (the consuming GEMM's K loop carries it on to global memory)
        else if (row < M) {
          float *dst = S.dz + (long long)row * 256 + c0;
          *reinterpret_cast<gf4 *>(dst) = gf4{o[0], o[1], o[2], o[3]}; *reinterpret_cast<gf4 *>(dst + 4) = gf4{o[4], o[5], o[6], o[7]};
        }
      }
    }
    CHAIN_STAMP(stamp++);
    if (!more) break;
    zero();
    if (!PRE) G.early();
    G.late(); G.loop(acc);
    CHAIN_STAMP(stamp++);
  }
#undef DY
  if (DX) {
    gf4 accx[MT][1];
#pragma unroll
    for (int a = 0; a < MT; a++) accx[a][0] = gf4{0.f, 0.f, 0.f, 0.f};
    if (MT == 5 && EPI == 2) Gx.early();
    Gx.late(); Gx.loop(accx);
    CHAIN_STAMP(stamp++);
    // native tile layout (k_gemm_act's generic store): register r = C[16 a + 4 kq + r][16 wave + li]
    const int col = wave * 16 + li;
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * a + 4 * kq + r;
        if (row < M && col < P.dx_cols) P.dx[(long long)row * P.lddx + col] = accx[a][0][r] + 0.f;
      }
  }
  CHAIN_STAMP(stamp++);
}
