// csrc/wave_matvec.h — the tree-sparse mat-vecs of the solver stage on the matrix cores (chain layout; included by wave_physics.h).
// EXPERIMENT, OFF BY DEFAULT (-DTMW_MFMA_MATVEC): measured slower than the vector-ALU mat-vecs it would replace (round 4, below).
//
// out = A x over the ANCESTORS of every dof (rows of the ancestor-sparse storage: N s, the row half of M x) and out = A^T x over its
// DESCENDANTS (columns: D^-1 N^T g, the column half of M x) are 1 046 multiply-adds each on a 73-dof tree.  The vector-ALU versions
// (tmw_row_runs2: lane = dof, 55 masked steps; tmw_rows_colacc: lane = depth, 73 row steps) use 14 - 19 of 64 lanes per step and are the
// two most expensive primitives of a CG iteration (6.6 % + 6.1 % of a substep, profiles/r03_phase_profile.txt).  Here the dofs are cut into six
// TILES of <= 16 rows (trunk; two 8-dof limbs; the 9-dof limb; the tail in two pieces; two 6-dof limbs) and a product is a handful of
// v_mfma_f32_16x16x4_f32 per tile: the matrix entries of FOUR ancestor columns x 16 rows (resp. four descendant rows x 16 columns) come
// from LDS with ONE ds_read_b32 per instruction straight in operand layout — the address is `row end - depth of the column`, affine in the
// lane — the vector operand is a 4-address broadcast read, and the entries outside the tree's sparsity are cut by compile-time lane masks:
// 37 + 39 matrix instructions per pair of products; fp32 in, fp32 accumulate (a k-ordered fmaf chain: same precision, another order).
//
// MEASURED (MI355X, 4096 envs, in-kernel phase profile, cycles per substep; env.step of one launch): vector-ALU N s 21.5 k, D^-1 N^T g 20.1 k,
// env.step 2.76 ms.  First version here (tile after tile, every matrix instruction behind its own LDS round trip and its predecessor's 40-cycle
// accumulator latency): 21.7 k / 38.3 k, 3.09 ms.  This version (all operand loads of three tiles first, then the matrix instructions round-robin
// over their accumulators): 18.2 k / 26.0 k, 3.01 ms — and the OTHER phases of the substep slow down by more than these two gain: an fp32 MFMA
// has the vector ALU's multiply-add rate (a mat-vec uses 1 / 16 of its columns), holds the SIMD's issue for 8 of its 32 cycles, and three
// waves per SIMD doing this share one matrix pipe.  A 73-dof mat-vec is too small and too latency-bound for the matrix cores; the 12 x 61 x 12
// Schur block and the trunk products of the factorisation (wave_physics.h), with sixteen useful columns, are where they pay.
//
// Layout facts used (wave_layout.h): row i of the storage is [A(i,i), A(i,parent), .., A(i,root)], i.e. the entry of the ancestor at depth q
// sits at  E_i - q  with  E_i = Madr_i + depth_i  (TMW_MEND of the dof's packed table word); the dofs of a chain are consecutive and so are
// their depths; the trunk's dof j has depth j.
#pragma once

struct TmwSeg { int first, n, row0, cfirst, clast, d0; };     // rows [first, first + n) of the chain [cfirst, clast] (hanging off trunk depth d0) at tile rows row0 ..
template <int T> struct TmwTile;
template <> struct TmwTile<0> { static constexpr int NS = 1; static constexpr TmwSeg S[2] = {{0, 12, 0, 0, 11, 0}, {0, 0, 0, 0, 0, 0}}; };
template <> struct TmwTile<1> { static constexpr int NS = 2; static constexpr TmwSeg S[2] = {{65, 8, 0, 65, 72, 6}, {57, 8, 8, 57, 64, 6}}; };
template <> struct TmwTile<2> { static constexpr int NS = 1; static constexpr TmwSeg S[2] = {{48, 9, 0, 48, 56, 6}, {0, 0, 0, 0, 0, 0}}; };
template <> struct TmwTile<3> { static constexpr int NS = 1; static constexpr TmwSeg S[2] = {{24, 16, 0, 24, 47, 12}, {0, 0, 0, 0, 0, 0}}; };
template <> struct TmwTile<4> { static constexpr int NS = 1; static constexpr TmwSeg S[2] = {{40, 8, 0, 24, 47, 12}, {0, 0, 0, 0, 0, 0}}; };
template <> struct TmwTile<5> { static constexpr int NS = 2; static constexpr TmwSeg S[2] = {{18, 6, 0, 18, 23, 12}, {12, 6, 8, 12, 17, 12}}; };
#define TMW_NTILE 6

constexpr int tmw_mv_parent(int i) {
  if (i <= 0) return -1;
  if (i < TMW_RODENT_TRUNK) return i - 1;
  return tmw_chain_first(i) == i ? tmw_chain_depth(i) - 1 : i - 1;      // a leaf chain hangs off the trunk dof one level up
}
constexpr bool tmw_mv_is_anc(int col, int row, bool self) {      // col is an ancestor of row (or row itself, if `self`)
  if (col == row) return self;
  for (int r = tmw_mv_parent(row); r >= 0; r = tmw_mv_parent(r)) if (r == col) return true;
  return false;
}
template <int T> constexpr int tmw_tile_dof(int il) {             // dof of tile row il, -1 = none
  for (int s = 0; s < TmwTile<T>::NS; s++) {
    const TmwSeg g = TmwTile<T>::S[s];
    if (il >= g.row0 && il < g.row0 + g.n) return g.first + il - g.row0;
  }
  return -1;
}
// every dof belongs to exactly one tile row, and the segments agree with the chain table of wave_layout.h
constexpr bool tmw_tiles_cover() {
  int seen[73] = {};
  for (int il = 0; il < 16; il++) {
    const int d[TMW_NTILE] = {tmw_tile_dof<0>(il), tmw_tile_dof<1>(il), tmw_tile_dof<2>(il), tmw_tile_dof<3>(il), tmw_tile_dof<4>(il), tmw_tile_dof<5>(il)};
    for (int t = 0; t < TMW_NTILE; t++) if (d[t] >= 0) { if (d[t] >= 73) return false; seen[d[t]]++; }
  }
  for (int i = 0; i < 73; i++) if (seen[i] != 1) return false;
  return true;
}
template <int T> constexpr bool tmw_tile_consistent() {
  for (int s = 0; s < TmwTile<T>::NS; s++) {
    const TmwSeg g = TmwTile<T>::S[s];
    if (tmw_chain_first(g.first) != g.cfirst || tmw_chain_first(g.clast) != g.cfirst || (g.clast < 72 && tmw_chain_first(g.clast + 1) == g.cfirst && g.cfirst != 0)) return false;
    if (g.cfirst == 0 ? (g.d0 != 0 || g.clast != TMW_RODENT_TRUNK - 1) : tmw_chain_depth(g.cfirst) != g.d0) return false;
    if ((g.row0 & 3) != 0 || g.first + g.n - 1 > g.clast) return false;       // (output rows are written in groups of four consecutive dofs)
  }
  return true;
}
static_assert(tmw_tiles_cover(), "the mat-vec tiles must cover every dof once");
static_assert(tmw_tile_consistent<0>() && tmw_tile_consistent<1>() && tmw_tile_consistent<2>() && tmw_tile_consistent<3>() && tmw_tile_consistent<4>() && tmw_tile_consistent<5>(),
              "the mat-vec tiles must agree with TMW_RODENT_LEAF_CHAINS");

struct TmwSteps { int n; int c0[24]; };
// ROW product of tile T: blocks of four ancestor COLUMNS — the trunk ancestors first, then each segment's own chain up to the tile's last row
template <int T> constexpr TmwSteps tmw_row_steps() {
  TmwSteps r = {0, {}};
  const int ntr = T == 0 ? TMW_RODENT_TRUNK : TmwTile<T>::S[0].d0;
  for (int c = 0; c < ntr; c += 4) r.c0[r.n++] = c;
  if (T != 0)
    for (int s = 0; s < TmwTile<T>::NS; s++) {
      const TmwSeg g = TmwTile<T>::S[s];
      for (int c = g.cfirst; c <= g.first + g.n - 1; c += 4) r.c0[r.n++] = c;
    }
  return r;
}
// COLUMN product of tile T: blocks of four descendant ROWS — for the trunk tile every chain (and the trunk itself), else the rest of the
// segments' chains from the tile's first row on
template <int T> constexpr TmwSteps tmw_col_steps() {
  TmwSteps r = {0, {}};
  if (T == 0) {
#define TMW_X(first_, len_, d0_) for (int c = first_; c < first_ + len_; c += 4) r.c0[r.n++] = c;
    TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
    for (int c = 0; c < TMW_RODENT_TRUNK; c += 4) r.c0[r.n++] = c;
  } else
    for (int s = 0; s < TmwTile<T>::NS; s++) {
      const TmwSeg g = TmwTile<T>::S[s];
      for (int c = g.first; c <= g.clast; c += 4) r.c0[r.n++] = c;
    }
  return r;
}
// lane mask of a row-product step: lane 16 kk + il carries A(row il of the tile, column c0 + kk)
template <int T, bool DIAG> constexpr unsigned long long tmw_row_mask(int c0) {
  unsigned long long m = 0;
  for (int kk = 0; kk < 4; kk++)
    for (int il = 0; il < 16; il++) {
      const int row = tmw_tile_dof<T>(il), col = c0 + kk;
      if (row >= 0 && col < 73 && tmw_chain_first(col) == tmw_chain_first(c0) && tmw_mv_is_anc(col, row, DIAG)) m |= 1ull << (16 * kk + il);
    }
  return m;
}
// ... of a column-product step: lane 16 kk + j carries A(row r0 + kk, column j of the tile)
template <int T> constexpr unsigned long long tmw_col_mask(int r0) {
  unsigned long long m = 0;
  for (int kk = 0; kk < 4; kk++)
    for (int j = 0; j < 16; j++) {
      const int col = tmw_tile_dof<T>(j), row = r0 + kk;
      if (col >= 0 && row < 73 && tmw_chain_first(row) == tmw_chain_first(r0) && tmw_mv_is_anc(col, row, false)) m |= 1ull << (16 * kk + j);
    }
  return m;
}
template <int T> constexpr int tmw_row_dmax() { const TmwSteps s = tmw_row_steps<T>(); int d = 0; for (int i = 0; i < s.n; i++) if (tmw_chain_depth(s.c0[i]) > d) d = tmw_chain_depth(s.c0[i]); return d; }
template <int T> constexpr unsigned long long tmw_tile_rows4_mask(int r) {       // lanes 16 b (one per 16-lane block) whose row 4 b + r exists
  unsigned long long m = 0;
  for (int b = 0; b < 4; b++) if (tmw_tile_dof<T>(4 * b + r) >= 0) m |= 1ull << (16 * b);
  return m;
}
template <int T> constexpr unsigned long long tmw_tile_store_mask() {            // lanes 16 b + j whose row 4 b + (j & 3) exists
  unsigned long long m = 0;
  for (int b = 0; b < 4; b++) for (int j = 0; j < 16; j++) if (tmw_tile_dof<T>(4 * b + (j & 3)) >= 0) m |= 1ull << (16 * b + j);
  return m;
}
template <int T> constexpr unsigned long long tmw_tile_cols_mask() { unsigned long long m = 0; for (int j = 0; j < 16; j++) if (tmw_tile_dof<T>(j) >= 0) m |= 1ull << j; return m; }
// E_k of the rows of a chain grows quadratically: E(k + 4) - E(k) = 4 (d0 + 1) + 4 k + 10 for the k-th dof of a chain hanging off depth d0
constexpr int tmw_mv_end(int dof) { return tmw_chain_madr(dof) + tmw_chain_depth(dof); }

// D[i][j] += sum_kk A[i][kk] B[kk][j], operands one value per lane (A: lane 16 kk + i, B: lane 16 kk + j), D[i][j] in acc[i % 4] lane 16 (i / 4) + j
#ifdef TM_HOST_EMU
TM_DEV void tmw_mma(const float *a, const float *b, float (*acc)[TMW_NL]) {
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float s = acc[i & 3][16 * (i >> 2) + j];
      for (int kk = 0; kk < 4; kk++) s = fmaf(a[16 * kk + i], b[16 * kk + j], s);
      acc[i & 3][16 * (i >> 2) + j] = s;
    }
}
#else
TM_DEV void tmw_mma(const float *a, const float *b, float (*acc)[TMW_NL]) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  f4 v = {acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
  v = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], v, 0, 0, 0);
  acc[0][0] = v[0]; acc[1][0] = v[1]; acc[2][0] = v[2]; acc[3][0] = v[3];
}
#endif
template <int... I> struct tmw_seq {};
template <int N, int... I> struct tmw_make_seq : tmw_make_seq<N - 1, N - 1, I...> {};
template <int... I> struct tmw_make_seq<0, I...> { typedef tmw_seq<I...> type; };
template <int LO, int N, int... I> struct tmw_make_range : tmw_make_range<LO, N - 1, LO + N - 1, I...> {};
template <int LO, int... I> struct tmw_make_range<LO, 0, I...> { typedef tmw_seq<I...> type; };

// dof of tile row il for this lane (run time; rows the tile does not have map to some valid dof and are masked)
template <int T> TM_DEV int tmw_lane_tile_dof(int il) {
  constexpr int ns = TmwTile<T>::NS, f0 = TmwTile<T>::S[0].first, f1 = TmwTile<T>::S[1].first, r1 = TmwTile<T>::S[1].row0;
  if (ns == 2) return il < r1 ? f0 + il : f1 + il - r1;
  return f0 + il;
}

// Execution order (what made the first version of this file SLOWER than the vector-ALU mat-vecs it replaces: one tile after the other, each
// instruction behind its own LDS round trip and its predecessor's 40-cycle accumulator latency — 130 - 180 cycles per matrix instruction):
//   phase 1  every operand load of ALL six tiles is issued (one ds_read_b32 per matrix operand, one per vector operand),
//   phase 2  the matrix operands are masked,
//   phase 3  the matrix instructions run ROUND-ROBIN over the six tiles' accumulators (consecutive instructions never depend on each other),
//   phase 4  one store per tile.
// `__builtin_amdgcn_sched_barrier` keeps the compiler from folding the phases back into load -> use chains.
#ifdef TM_HOST_EMU
#define TMW_MV_FENCE() do { } while (0)
#else
#define TMW_MV_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define TMW_MV_MAXSTEP 24
// (sized exactly: a struct with 24-entry arrays stayed in scratch memory at one of its call sites)
template <int T, int NM, int OFF = 0> struct TmwMvOpsN { static constexpr int off = OFF; float m[NM][TMW_NL], v[NM][TMW_NL], acc[4][TMW_NL]; int ab[TMW_NL], ob[TMW_NL]; };

// ---- rows: out_i = sum over the ancestors a of i (and i itself if DIAG) of A(i, a) x_a.  out may not alias x.
template <int T, bool DIAG, int S, typename OPS>
TM_DEV void tmw_rowprod_load(float *L, OPS &o, const int *xb) {
  constexpr TmwSteps st = tmw_row_steps<T>();
  constexpr int c0 = st.c0[S], off = tmw_row_dmax<T>() - tmw_chain_depth(c0);
#ifdef TM_HOST_EMU
  for (int lane = 0; lane < 64; lane++) { o.m[S][lane] = L[o.ab[lane] + off]; o.v[S][lane] = L[xb[lane] + c0]; }
#else
  o.m[S][0] = L[o.ab[0] + off]; o.v[S][0] = L[xb[0] + c0];
#endif
}
template <int T, bool DIAG, int S, typename OPS>
TM_DEV void tmw_rowprod_mask(OPS &o) {
  constexpr TmwSteps st = tmw_row_steps<T>();
  constexpr unsigned long long mask = tmw_row_mask<T, DIAG>(st.c0[S]);
#ifdef TM_HOST_EMU
  for (int lane = 0; lane < 64; lane++) o.m[S][lane] = TMW_MASK(mask) ? o.m[S][lane] : 0.f;
#else
  o.m[S][0] = TMW_MASK(mask) ? o.m[S][0] : 0.f;
#endif
}
template <int T, bool DIAG, typename OPS, int... S> TM_DEV void tmw_rowprod_loads(float *L, OPS &o, const int *xb, tmw_seq<S...>) { int d[] = {0, (tmw_rowprod_load<T, DIAG, S>(L, o, xb), 0)...}; (void)d; }
template <int T, bool DIAG, typename OPS, int... S> TM_DEV void tmw_rowprod_masks(OPS &o, tmw_seq<S...>) { int d[] = {0, (tmw_rowprod_mask<T, DIAG, S>(o), 0)...}; (void)d; }
template <int T, int NSTEP, int S, typename OPS> TM_DEV void tmw_mv_mma_step(OPS &o, bool vec_first) {
  if (S < NSTEP) { if (vec_first) tmw_mma(o.v[S < NSTEP ? S : 0], o.m[S < NSTEP ? S : 0], o.acc); else tmw_mma(o.m[S < NSTEP ? S : 0], o.v[S < NSTEP ? S : 0], o.acc); }
}
template <int T, bool DIAG, typename OPS>
TM_DEV void tmw_rowprod_setup(WCtx &c, const WLayout &K, int A, int out, OPS &o) {
  float *L = c.L; TMW_LANE_DECL
  constexpr int DMAX = tmw_row_dmax<T>();
  TMW_FOR {
    const int il = lane & 15, kk = lane >> 4;
    const int dof = tmw_lane_tile_dof<T>(il);
    o.ab[TMW_LI] = A + TMW_MEND(TMW_W0(dof)) - kk - DMAX;                 // the step adds DMAX - depth(its first column); the lane's kk is in
    o.ob[TMW_LI] = out + tmw_lane_tile_dof<T>(4 * kk) + (il & 3);       // lane 16 b + j stores row 4 b + (j & 3): four lanes per row, same word, same value
    for (int r = 0; r < 4; r++) o.acc[r][TMW_LI] = 0.f;
  }
}
// D[i][j] is the same for every j: lane 16 b + j takes row 4 b + (j & 3) out of component j & 3 — ONE store per tile
template <int T, typename OPS>
TM_DEV void tmw_rowprod_store(WCtx &c, OPS &o) {
  float *L = c.L; TMW_LANE_DECL
  TMW_FOR {
    // (compile-time lane masks, not `acc[lane & 3]`-style compares: the optimiser turns a compare chain over the components into a dynamically
    // indexed load and the whole operand struct then lives in scratch memory)
    float a0 = o.acc[0][TMW_LI], a1 = o.acc[1][TMW_LI], a2 = o.acc[2][TMW_LI], v = o.acc[3][TMW_LI];
    v = TMW_MASK(0x4444444444444444ull) ? a2 : v;
    v = TMW_MASK(0x2222222222222222ull) ? a1 : v;
    v = TMW_MASK(0x1111111111111111ull) ? a0 : v;
    constexpr unsigned long long sm = tmw_tile_store_mask<T>();
    if (sm == ~0ull) L[o.ob[TMW_LI]] = v;
    else if (TMW_MASK(sm)) L[o.ob[TMW_LI]] = v;
  }
}
#define TMW_MV_TILES(X) X(0) X(1) X(2) X(3) X(4) X(5)
// three tiles at a time (all six: 74 operand registers on top of the solver's own — the allocator spilled 26 of them)
template <bool DIAG, int TA, int TB, int TC>
TM_DEV void tmw_rowprod_group(WCtx &c, const WLayout &K, int A, int out, const int *xb) {
  float *L = c.L;
  constexpr int NA = tmw_row_steps<TA>().n, NB = tmw_row_steps<TB>().n, NC = tmw_row_steps<TC>().n;
  TmwMvOpsN<TA, NA> oa; TmwMvOpsN<TB, NB> ob; TmwMvOpsN<TC, NC> oc;
  static_assert(NA <= 10 && NB <= 10 && NC <= 10, "row steps");
  tmw_rowprod_setup<TA, DIAG>(c, K, A, out, oa); tmw_rowprod_setup<TB, DIAG>(c, K, A, out, ob); tmw_rowprod_setup<TC, DIAG>(c, K, A, out, oc);
  TMW_MV_FENCE();
  tmw_rowprod_loads<TA, DIAG>(L, oa, xb, typename tmw_make_seq<NA>::type()); tmw_rowprod_loads<TB, DIAG>(L, ob, xb, typename tmw_make_seq<NB>::type());
  tmw_rowprod_loads<TC, DIAG>(L, oc, xb, typename tmw_make_seq<NC>::type());
  TMW_MV_FENCE();
  tmw_rowprod_masks<TA, DIAG>(oa, typename tmw_make_seq<NA>::type()); tmw_rowprod_masks<TB, DIAG>(ob, typename tmw_make_seq<NB>::type());
  tmw_rowprod_masks<TC, DIAG>(oc, typename tmw_make_seq<NC>::type());
  TMW_MV_FENCE();
#define TMW_STEP(S) tmw_mv_mma_step<TA, NA, S>(oa, false); tmw_mv_mma_step<TB, NB, S>(ob, false); tmw_mv_mma_step<TC, NC, S>(oc, false);
  TMW_STEP(0) TMW_STEP(1) TMW_STEP(2) TMW_STEP(3) TMW_STEP(4) TMW_STEP(5) TMW_STEP(6) TMW_STEP(7) TMW_STEP(8) TMW_STEP(9)
#undef TMW_STEP
  TMW_MV_FENCE();
  tmw_rowprod_store<TA>(c, oa); tmw_rowprod_store<TB>(c, ob); tmw_rowprod_store<TC>(c, oc);
}
template <bool DIAG>
TM_DEV void tmw_rowprod_mfma(WCtx &c, const WLayout &K, int A, int x, int out) {
  TMW_LANE_DECL
  TMW_REG(int, xb);
  TMW_FOR { xb[TMW_LI] = x + (lane >> 4); }
  tmw_rowprod_group<DIAG, 4, 0, 2>(c, K, A, out, xb);       // 9 + 3 + 5 matrix instructions
  tmw_rowprod_group<DIAG, 3, 1, 5>(c, K, A, out, xb);       // 7 + 6 + 7
  TMW_SYNC();
}

// ---- columns: c_j = sum over the descendants i of j of A(i, j) x_i.
//   MODE 0: out_j = c_j;   MODE 1: out_j = (x_j + c_j) * Dinv_j  (D^-1 N^T x with A = N);   MODE 2: out_j += c_j
// out may alias x: every x is read (phase 1 of both rounds) before the first store.
template <int T> struct TmwColAddr { int ea[TMW_NL], dc[TMW_NL], cd[TMW_NL]; };
template <int T, int S> constexpr bool tmw_col_chain_start() {
  constexpr TmwSteps st = tmw_col_steps<T>();
  return S == 0 || tmw_chain_first(st.c0[S > 0 ? S - 1 : 0]) != tmw_chain_first(st.c0[S]) || st.c0[S > 0 ? S - 1 : 0] + 4 != st.c0[S];
}
// phase 0: E of row r0 + kk at the start of every chain's rows, from the dof table (one read each, issued together)
template <int T, int S, typename OPS>
TM_DEV void tmw_colprod_estart(float *L, OPS &o, const int *tb) {
  constexpr TmwSteps st = tmw_col_steps<T>();
  constexpr int r0 = st.c0[S];
  if (tmw_col_chain_start<T, S>()) {
#ifdef TM_HOST_EMU
    for (int lane = 0; lane < 64; lane++) o.m[S - OPS::off][lane] = L[tb[lane] + 2 * r0];
#else
    o.m[S - OPS::off][0] = L[tb[0] + 2 * r0];
#endif
  }
}
// phase 1: the address of step S (read at a chain start, else advanced by the closed form E(k + 4) - E(k) = 4 d0 + 4 k + 14 for the k-th dof of
// a chain whose first dof has depth d0 — checked against the layout) and the two operand loads
template <int T, int S, typename OPS>
TM_DEV void tmw_colprod_load(float *L, OPS &o, TmwColAddr<T> &ad, const int *xb, const int *kk4) {
  constexpr TmwSteps st = tmw_col_steps<T>();
  constexpr int r0 = st.c0[S];
  constexpr bool chain_start = tmw_col_chain_start<T, S>();
  constexpr int k0 = r0 - 4 - tmw_chain_first(r0), d0 = tmw_chain_depth(tmw_chain_first(r0));
  constexpr int inc = 4 * d0 + 4 * k0 + 14;
  static_assert(chain_start || (tmw_mv_end(r0) - tmw_mv_end(r0 - 4) == inc && (r0 + 1 > 72 || tmw_chain_first(r0 + 1) != tmw_chain_first(r0) || tmw_mv_end(r0 + 1) - tmw_mv_end(r0 - 3) == inc + 4)),
                "closed form of the row-end increments");
#ifdef TM_HOST_EMU
  for (int lane = 0; lane < 64; lane++) {
    if (chain_start) ad.ea[lane] = TMW_MEND(tm_f2i(o.m[S - OPS::off][lane])); else ad.ea[lane] += inc + kk4[lane];
    o.m[S - OPS::off][lane] = L[ad.ea[lane] - ad.dc[lane]]; o.v[S - OPS::off][lane] = L[xb[lane] + r0];
  }
#else
  if (chain_start) ad.ea[0] = TMW_MEND(tm_f2i(o.m[S - OPS::off][0])); else ad.ea[0] += inc + kk4[0];
  o.m[S - OPS::off][0] = L[ad.ea[0] - ad.dc[0]]; o.v[S - OPS::off][0] = L[xb[0] + r0];
#endif
}
template <int T, int S, typename OPS>
TM_DEV void tmw_colprod_mask(OPS &o) {
  constexpr TmwSteps st = tmw_col_steps<T>();
  constexpr unsigned long long mask = tmw_col_mask<T>(st.c0[S]);
#ifdef TM_HOST_EMU
  for (int lane = 0; lane < 64; lane++) o.m[S - OPS::off][lane] = TMW_MASK(mask) ? o.m[S - OPS::off][lane] : 0.f;
#else
  o.m[S - OPS::off][0] = TMW_MASK(mask) ? o.m[S - OPS::off][0] : 0.f;
#endif
}
template <int T, typename OPS, int... S> TM_DEV void tmw_colprod_estarts(float *L, OPS &o, const int *tb, tmw_seq<S...>) { int d[] = {0, (tmw_colprod_estart<T, S>(L, o, tb), 0)...}; (void)d; }
template <int T, typename OPS, int... S> TM_DEV void tmw_colprod_loads(float *L, OPS &o, TmwColAddr<T> &ad, const int *xb, const int *kk4, tmw_seq<S...>) { int d[] = {0, (tmw_colprod_load<T, S>(L, o, ad, xb, kk4), 0)...}; (void)d; }
template <int T, typename OPS, int... S> TM_DEV void tmw_colprod_masks(OPS &o, tmw_seq<S...>) { int d[] = {0, (tmw_colprod_mask<T, S>(o), 0)...}; (void)d; }
template <int T, typename OPS>
TM_DEV void tmw_colprod_setup(WCtx &c, const WLayout &K, int A, OPS &o, TmwColAddr<T> &ad) {
  float *L = c.L; TMW_LANE_DECL
  TMW_FOR {
    const int dof = tmw_lane_tile_dof<T>(lane & 15);
    ad.cd[TMW_LI] = dof;
    ad.dc[TMW_LI] = TMW_DEPTH(TMW_W0(dof)) - A;                  // entry (row, col j) at  A + E_row - depth_j
    ad.ea[TMW_LI] = 0;
    for (int r = 0; r < 4; r++) o.acc[r][TMW_LI] = 0.f;
  }
}
template <int T, int MODE, typename OPS>
TM_DEV void tmw_colprod_store(WCtx &c, const WLayout &K, int x, int out, OPS &o, TmwColAddr<T> &ad) {
  float *L = c.L; TMW_LANE_DECL
  TMW_FOR {
    if (TMW_MASK(tmw_tile_cols_mask<T>())) {                    // row 0 of D: lanes 0 .. 15 of acc[0] = the 16 columns
      const int dof = ad.cd[TMW_LI];
      float v = o.acc[0][TMW_LI];
      if (MODE == 1) v = (L[x + dof] + v) * L[K.l_Dinv + dof];
      if (MODE == 2) v += L[out + dof];
      L[out + dof] = v;
    }
  }
}
// steps [LO, LO + N) of tile T through phases 0 .. 2
template <int T, int LO, int N, typename OPS> TM_DEV void tmw_colprod_p0(float *L, OPS &o, const int *tb) { tmw_colprod_estarts<T>(L, o, tb, typename tmw_make_range<LO, N>::type()); }
template <int T, int LO, int N, typename OPS> TM_DEV void tmw_colprod_p1(float *L, OPS &o, TmwColAddr<T> &ad, const int *xb, const int *kk4) { tmw_colprod_loads<T>(L, o, ad, xb, kk4, typename tmw_make_range<LO, N>::type()); }
template <int T, int LO, int N, typename OPS> TM_DEV void tmw_colprod_p2(OPS &o) { tmw_colprod_masks<T>(o, typename tmw_make_range<LO, N>::type()); }
template <int MODE>
TM_DEV void tmw_colprod_mfma(WCtx &c, const WLayout &K, int A, int x, int out) {
  float *L = c.L; TMW_LANE_DECL
  TMW_REG(int, xb); TMW_REG(int, tb); TMW_REG(int, kk4);
  TMW_FOR { const int kk = lane >> 4; xb[TMW_LI] = x + kk; tb[TMW_LI] = K.l_tdof + 2 * kk; kk4[TMW_LI] = 4 * kk; }
  // The trunk tile has 20 steps (every chain's rows), the others 2 .. 6.  Two rounds of at most ~ 20 instructions keep the operand registers
  // bounded; the trunk tile's chain is cut in two halves on TWO accumulators (o0, and o0b for steps 10 ..), added in the store
  constexpr int N0 = tmw_col_steps<0>().n, H0 = N0 / 2;
  constexpr int N1 = tmw_col_steps<1>().n, N2 = tmw_col_steps<2>().n, N3 = tmw_col_steps<3>().n, N4 = tmw_col_steps<4>().n, N5 = tmw_col_steps<5>().n;
  static_assert(N0 <= 24 && H0 <= 12 && N0 - H0 <= 12 && N1 <= 12 && N2 <= 12 && N3 <= 12 && N4 <= 12 && N5 <= 12, "column steps");
  TmwMvOpsN<0, H0> o0; TmwColAddr<0> ad0; tmw_colprod_setup<0>(c, K, A, o0, ad0);
  TmwMvOpsN<0, N0 - H0, H0> o0b;
  TMW_FOR { for (int r = 0; r < 4; r++) o0b.acc[r][TMW_LI] = 0.f; }
  TmwMvOpsN<3, N3> o3; TmwColAddr<3> ad3; tmw_colprod_setup<3>(c, K, A, o3, ad3);
  TmwMvOpsN<2, N2> o2; TmwColAddr<2> ad2; tmw_colprod_setup<2>(c, K, A, o2, ad2);
  {   // round 1: trunk tile steps [0, H0), tiles 3 and 2 (their stores wait for round 2: with out == x its loads still read their columns' x)
    tmw_colprod_p0<0, 0, H0>(L, o0, tb); tmw_colprod_p0<3, 0, N3>(L, o3, tb); tmw_colprod_p0<2, 0, N2>(L, o2, tb);
    TMW_MV_FENCE();
    tmw_colprod_p1<0, 0, H0>(L, o0, ad0, xb, kk4); tmw_colprod_p1<3, 0, N3>(L, o3, ad3, xb, kk4); tmw_colprod_p1<2, 0, N2>(L, o2, ad2, xb, kk4);
    TMW_MV_FENCE();
    tmw_colprod_p2<0, 0, H0>(o0); tmw_colprod_p2<3, 0, N3>(o3); tmw_colprod_p2<2, 0, N2>(o2);
    TMW_MV_FENCE();
#define TMW_STEP(S) tmw_mv_mma_step<0, H0, S>(o0, true); tmw_mv_mma_step<3, N3, S>(o3, true); tmw_mv_mma_step<2, N2, S>(o2, true);
    TMW_STEP(0) TMW_STEP(1) TMW_STEP(2) TMW_STEP(3) TMW_STEP(4) TMW_STEP(5) TMW_STEP(6) TMW_STEP(7) TMW_STEP(8) TMW_STEP(9) TMW_STEP(10) TMW_STEP(11)
#undef TMW_STEP
    TMW_MV_FENCE();
  }
  {   // round 2: trunk tile steps [H0, N0) on the second accumulator, tiles 1, 5, 4
    TmwMvOpsN<1, N1> o1; TmwColAddr<1> ad1; tmw_colprod_setup<1>(c, K, A, o1, ad1);
    TmwMvOpsN<5, N5> o5; TmwColAddr<5> ad5; tmw_colprod_setup<5>(c, K, A, o5, ad5);
    TmwMvOpsN<4, N4> o4; TmwColAddr<4> ad4; tmw_colprod_setup<4>(c, K, A, o4, ad4);
    tmw_colprod_p0<0, H0, N0 - H0>(L, o0b, tb); tmw_colprod_p0<1, 0, N1>(L, o1, tb); tmw_colprod_p0<5, 0, N5>(L, o5, tb); tmw_colprod_p0<4, 0, N4>(L, o4, tb);
    TMW_MV_FENCE();
    tmw_colprod_p1<0, H0, N0 - H0>(L, o0b, ad0, xb, kk4); tmw_colprod_p1<1, 0, N1>(L, o1, ad1, xb, kk4); tmw_colprod_p1<5, 0, N5>(L, o5, ad5, xb, kk4); tmw_colprod_p1<4, 0, N4>(L, o4, ad4, xb, kk4);
    TMW_MV_FENCE();
    tmw_colprod_p2<0, H0, N0 - H0>(o0b); tmw_colprod_p2<1, 0, N1>(o1); tmw_colprod_p2<5, 0, N5>(o5); tmw_colprod_p2<4, 0, N4>(o4);
    TMW_MV_FENCE();
#define TMW_STEP(S) tmw_mv_mma_step<0, N0 - H0, S>(o0b, true); tmw_mv_mma_step<1, N1, S>(o1, true); tmw_mv_mma_step<5, N5, S>(o5, true); tmw_mv_mma_step<4, N4, S>(o4, true);
    TMW_STEP(0) TMW_STEP(1) TMW_STEP(2) TMW_STEP(3) TMW_STEP(4) TMW_STEP(5) TMW_STEP(6) TMW_STEP(7) TMW_STEP(8) TMW_STEP(9) TMW_STEP(10) TMW_STEP(11)
#undef TMW_STEP
    TMW_MV_FENCE();
    TMW_FOR { o0.acc[0][TMW_LI] += o0b.acc[0][TMW_LI]; }          // (only row 0 of D is read)
    // (the trunk tile's store last within the round is fine: with out == x its columns are read by no later tile)
    tmw_colprod_store<0, MODE>(c, K, x, out, o0, ad0); tmw_colprod_store<3, MODE>(c, K, x, out, o3, ad3); tmw_colprod_store<2, MODE>(c, K, x, out, o2, ad2);
    tmw_colprod_store<1, MODE>(c, K, x, out, o1, ad1); tmw_colprod_store<5, MODE>(c, K, x, out, o5, ad5); tmw_colprod_store<4, MODE>(c, K, x, out, o4, ad4);
  }
  TMW_SYNC();
}
