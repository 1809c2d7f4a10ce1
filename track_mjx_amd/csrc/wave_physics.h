// csrc/wave_physics.h — K2, wave-per-env: one 64-lane wavefront integrates one env; every per-env array of a
// substep lives in LDS (wave_layout.h: 11 276 bytes per env for the rodent's chain layout = 9 LDS granules of 1280 B; the residency is set by the
// registers — ~130 VGPRs under __launch_bounds__(64, 3) => three waves per SIMD = 12 envs per CU; the inertia matrix, the warm start, qfrc_actuator
// and the activation state are additionally kept in a per-env global record between their uses), the 64 lanes split bodies /
// dofs / constraint rows / matrix columns between them.
//
// Same maths as tests/lane/physics_core.h, the lane-per-env cross-check build (MJX `mjx.step`, reference call site
// track_mjx/environment/task/single_clip_tracking.py:219) re-organised for the wavefront:
//   * kinematics and velocity/acceleration prefixes by POINTER JUMPING over the tree (log2(depth) = 6 rounds
//     instead of a 39-level walk); composite inertias / body forces by one branch-free reverse sweep,
//   * tree-sparse M; for the rodent's chain-structured dof tree the factorisation L^T D L, the inverse L^-1 (same ancestor
//     sparsity: every M^-1 apply of the CG loop is two sparse mat-vecs instead of two 73-step dependent sweeps), Euler's
//     (M + hD)^-1 b and the column halves of M x / M^-1 x run REGISTER-RESIDENT: lane = ancestor depth, a matrix row is
//     one register, updates are v_readlane + v_pk_fma_f32 (section "register-resident variants" below); any other tree
//     takes the LDS-resident left-looking path,
//   * matrix-free constraint Jacobian (spatial velocity per paw body, wrench accumulation per subset of paw bodies); only
//     the ACTIVE rows (violated limits, penetrating contacts) are numbered and enter the solver,
//   * the CG iteration runs in the coordinates y = L qacc of the factorisation (diagonal Gauss term and preconditioner: one split
//     M^-1 product per iteration, no product with M inside the loop),
//   * dot products / line-search sums by DPP wave reductions; the rows of a line search live in registers (<= 16 active rows:
//     the three candidates of an iteration side by side in three 16-lane rows).
//
// Single source for the GPU and for the TEST-ONLY host emulation: a block of lane code is written as
//   TMW_FOR { ... uses `lane` ... }  TMW_SYNC();
// On the GPU TMW_FOR is empty (the wave executes the block once, lane = threadIdx.x) and TMW_SYNC is a wave-level
// compiler barrier (one wave per workgroup: the LDS unit executes a wave's DS instructions in order); under TM_HOST_EMU
// TMW_FOR loops lane = 0..63.  Lane-private values that live across blocks are declared with TMW_REG (one slot on the
// GPU, 64 in emulation).
#pragma once
#include "tm_common.h"
#include "wave_layout.h"

#ifdef TM_HOST_EMU
#define TMW_NL 64
#define TMW_FOR for (int lane = 0; lane < 64; lane++)
#define TMW_SYNC() do { } while (0)
#define TMW_LI lane
#define TMW_LANE_DECL
#else
#define TMW_NL 1
#define TMW_FOR if (true)
// One wave per workgroup: the LDS unit executes a wave's DS instructions in issue order, so a ds_read issued after a
// ds_write of another lane of the SAME wave already sees the data — no s_waitcnt / s_barrier is needed, only a compiler
// barrier that keeps the program order of the LDS accesses (TMW_HARD_SYNC restores __syncthreads() for A/B checks).
#ifdef TMW_HARD_SYNC
#define TMW_SYNC() __syncthreads()
#else
#define TMW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#define TMW_LI 0
#define TMW_LANE_DECL const int lane = c.lane;
#endif
#define TMW_REG(type, name) type name[TMW_NL]
// may lanes that merely duplicate another lane's work also duplicate its (identical) LDS store?  Yes on the GPU (same word,
// same value, one instruction); not in the emulation, where the lanes of a block run one after the other
#ifdef TM_HOST_EMU
#define TMW_DUP_LANES_OK 0
#else
#define TMW_DUP_LANES_OK 1
#endif
#define TMW_DS 7   // floats per dof-scan entry: 6-vector + ancestor pointer

// The model is immutable for the whole launch.  Through a plain pointer the compiler cannot know that (the kernel stores to global memory),
// so every model scalar — iteration caps, tolerances, counts — became a VECTOR load with a `s_waitcnt vmcnt(0)` in the middle of the solver
// (214 vector loads, 1.7 scalar loads per wave-substep in profiles/sq_counters.json).  The constant address space says "never written":
// uniform reads become s_load (scalar cache), per-lane table reads invariant global loads that may move above stores.
#ifdef TM_HOST_EMU
typedef const DModel TmwModel;
#else
typedef const __attribute__((address_space(4))) DModel TmwModel;
#endif
// a small model array as a local (the vector helpers of tm_common.h take plain pointers)
#define TMW_LOCAL(name, n, src) float name[n]; { _Pragma("unroll") for (int k_ = 0; k_ < (n); k_++) name[k_] = (src)[k_]; }
#define TMW_EMU_HAS_QA 1      // (tests/hostemu: WCtx carries the qa* / ma* registers)
struct WCtx {
  TmwModel *mp;
  float *L;          // LDS
  float *st;         // global float state [rows][n]
  int n, e;
  int lane;
  unsigned long long *prof;   // TMW_PROFILE builds only: per-env phase cycle counters
  unsigned long long tlast;
  float *dump;                // tests only: lane-per-env workspace that receives intermediates (tmw_dump)
  int nact;                   // number of ACTIVE constraint rows of the current substep (compact row space, tmw_make_constraint)
  int nla;                    // ... of which violated joint limits (the active contacts' rows follow, four each)
  int rs;                     // 0: `st` is the [row][n_env] state; > 0: `st` is the env-major physics record with this stride
  // per-lane model constants of the J / J^T products, fetched once per launch (a global load in every call left its latency exposed):
  unsigned kmask[TMW_NL];     // lane = (subset, component) of tmw_jt_force: contact mask (slots 0..31) of the subset (valid if n_wsub * 6 <= 64)
  const float *action;        // [nu][n] action rows (direct mode, c.rs == 0) or null
  float qfs0[TMW_NL], qfs1[TMW_NL];   // lean layout: qfrc_smooth of dof lane / lane + 64 (tmw_velocity_inertia -> solver, Euler)
  float dg0[TMW_NL], dg1[TMW_NL];     // CG: D = 1 / Dinv of dof lane / lane + 64 (tmw_solve_cg; read lane-locally only)
  float wp0[TMW_NL], wp1[TMW_NL];     // CG: w = D^-1 ghat of the PREVIOUS gradient (Polak-Ribiere numerator)
  float qa0[TMW_NL], qa1[TMW_NL];     // lean layout, CG: the iterate qacc of dof lane / lane + 64 (in LDS only while a product with J or M reads it: l_qacc = l_mv)
  float ma0[TMW_NL], ma1[TMW_NL];     // lean layout, CG: ut = y - y_s (first M qacc; parked in / staged through l_Ma = l_Mgrad)
  int tp0[TMW_NL], tp1[TMW_NL];       // lean layout: packed index word of dof lane / lane + 64 (DModel::tpack; round 5: the table left LDS)
  float *mspill;              // chain layout (WLayout::m_spilled): this env's copy of M in global memory (nnz words, 64 readable words in front)
};
// solver statistics of the last substep, kept in the spare LDS word behind the centre of mass (registers are what this kernel has none
// left of): 64 * CG iterations (MJX data.solver_niter) + line-search iterations summed over them; tmw_dump unpacks it for the tests
#define TMW_STATS(K) L[(K).l_com + 3]
#if defined(TMW_PROFILE) && !defined(TM_HOST_EMU)
#define TMW_TICK(idx) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (c.prof && c.lane == 0) c.prof[idx] += t_ - c.tlast; c.tlast = t_; } while (0)
#define TMW_TICK2(idx) TMW_TICK(idx)
#define TMW_COUNT(idx, v) do { if (c.prof && c.lane == 0) c.prof[idx] += (v); } while (0)
#else
#define TMW_TICK(idx) do { } while (0)
#define TMW_TICK2(idx) do { } while (0)
#define TMW_COUNT(idx, v) do { } while (0)
#endif
// state word (row off + i) of this env: the C-ABI layout [row][n_env], or — when the launch goes through the env-major physics
// record (c.rs = record stride; tmjx_hip.hip: k_rec_in / k_rec_out) — word (off + i) of this env's contiguous record, so that the
// one-wavefront-per-env kernel moves full sectors instead of one 4-byte word per 32-byte sector
#define WST(off, i) c.st[c.rs ? (size_t)c.e * (size_t)c.rs + (size_t)((off) + (i)) : (size_t)((off) + (i)) * (size_t)c.n + (size_t)c.e]

// lean layout (K.lean, the rodent): the activation state stays in the env's global record, qfrc_smooth in two registers of the lane that
// produces and consumes it (dof i = lane, lane + 64: every loop over the dofs uses that mapping), the contact frames share the floor's normal (wave_layout.h); the generic layout keeps all three in LDS
#define TMW_ACT(a) (*(K.lean ? &WST(m.s_act, (a)) : &L[K.l_act + (a)]))
#define TMW_QFS(i) (K.lean ? ((i) < 64 ? c.qfs0[TMW_LI] : c.qfs1[TMW_LI]) : L[K.l_qfrc_smooth + (i)])
#define TMW_QFS_SET(i, v) do { if (!K.lean) L[K.l_qfrc_smooth + (i)] = (v); else if ((i) < 64) c.qfs0[TMW_LI] = (v); else c.qfs1[TMW_LI] = (v); } while (0)
// lean layout: the CG's iterate and ut = y - y_s are updated lane-locally by every pass of the loop — two registers per lane each (round 5); an
// LDS image exists only where a product needs the vector across lanes (J qacc, M qacc, D^-1 N^T ut at the start of the solve), staged through
// the words of the search direction / its gradient, which are dead then (wave_layout.h: l_qacc = l_mv, l_Ma = l_Mgrad)
#ifdef TMW_NO_QA_REGS
#define TMW_QAREG(K) 0
#else
#define TMW_QAREG(K) ((K).lean)
#endif
#define TMW_QA(i) (TMW_QAREG(K) ? ((i) < 64 ? c.qa0[TMW_LI] : c.qa1[TMW_LI]) : L[K.l_qacc + (i)])
#define TMW_QA_SET(i, v) do { if (!TMW_QAREG(K)) L[K.l_qacc + (i)] = (v); else if ((i) < 64) c.qa0[TMW_LI] = (v); else c.qa1[TMW_LI] = (v); } while (0)
#define TMW_MA(i) (TMW_QAREG(K) ? ((i) < 64 ? c.ma0[TMW_LI] : c.ma1[TMW_LI]) : L[K.l_Ma + (i)])
#define TMW_MA_SET(i, v) do { if (!TMW_QAREG(K)) L[K.l_Ma + (i)] = (v); else if ((i) < 64) c.ma0[TMW_LI] = (v); else c.ma1[TMW_LI] = (v); } while (0)
// Both dof slots of a lane (dofs lane and lane + 64) in ONE straight-line pass (round 6).  `for (i = lane; i < nv; i += 64)` compiles to a
// divergent two-trip loop: load, wait, use, branch, load, wait, use — two LDS round trips where the slots' loads can share one.  TMW_I1: an
// in-range LDS index for slot 1's loads (a lane without a second dof re-reads its slot-0 words; the results are selected away, never multiplied
// away: the words may hold anything).  Sums keep the loop's order and expressions (s += (ok ? a : 0) * b: the same contraction as the loop's s += a * b).
#define TMW_OK1(K) (lane + 64 < (K).nv)
#define TMW_I1(K) (TMW_OK1(K) ? lane + 64 : lane)
#define TMW_MA1(K, i1) (TMW_QAREG(K) ? c.ma1[TMW_LI] : L[(K).l_Ma + (i1)])
#define TMW_QA1(K, i1) (TMW_QAREG(K) ? c.qa1[TMW_LI] : L[(K).l_qacc + (i1)])
#define TMW_DG(i) ((i) < 64 ? c.dg0[TMW_LI] : c.dg1[TMW_LI])
#define TMW_WP(i) ((i) < 64 ? c.wp0[TMW_LI] : c.wp1[TMW_LI])
#define TMW_WP_SET(i, v) do { if ((i) < 64) c.wp0[TMW_LI] = (v); else c.wp1[TMW_LI] = (v); } while (0)
// friction coefficient of contact slot cc: lean layout = ONE coefficient for all slots (a uniform model read; tmjx_host::rodent_chains_match)
#define TMW_MU(cc) (K.lean ? m.con_mu[0] : L[K.l_con_mu + (cc)])
// efc_D of compact row kr: the four pyramid rows of a contact share ONE value (same penetration, impedance and weight), so the lean layout keeps
// it per contact — limits first (row = index), then one word per active contact (nlim + ncon words instead of nlim + 4 ncon; round 5)
#ifdef TMW_NO_EFCD_PACK
#define TMW_DIDX(kr) (kr)
#else
#define TMW_DIDX(kr) (K.lean ? ((kr) < c.nla ? (kr) : c.nla + (((kr) - c.nla) >> 2)) : (kr))
#endif
#define TMW_LIMSIGN(K) ((signed char *)(L + (K).l_lim_sign))      /* sign * (compact row + 1) of a violated limit, 0 otherwise */
TM_DEV int tm_f2i(float f) { int i; __builtin_memcpy(&i, &f, 4); return i; }
TM_DEV float tm_i2f(int i) { float f; __builtin_memcpy(&f, &i, 4); return f; }

// ---- wave reductions / broadcasts
#ifdef TM_HOST_EMU
TM_DEV float tmw_sum(const float *v) { float s = 0.f; for (int i = 0; i < 64; i++) s += v[i]; return s; }
TM_DEV float tmw_readlane(const float *v, int src) { return v[src]; }
#else
// (all steps with FULL row / bank masks: only the last lane of the reduced span is read afterwards, whose value does not depend on what the
// other lanes accumulate, and only then does the compiler fold the move into one `v_add_f32_dpp` — with the textbook partial masks each
// step was v_mov_b32_dpp + v_add_f32 + a re-zeroing v_mov)
template <int CTRL, int ROW_MASK, int BANK_MASK>
TM_DEV float tmw_dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, tm_f2i(v), CTRL, ROW_MASK, BANK_MASK, false);
  return v + tm_i2f(moved);
}
#ifdef TMW_SHFL_REDUCE
TM_DEV float tmw_sum(const float *vp) {
  float v = vp[0];
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
TM_DEV float tmw_readlane(const float *v, int src) { return __shfl(v[0], src); }
#define tmw_sum_dpp tmw_sum_unused
#define tmw_readlane_dpp tmw_readlane_unused
#else
#define tmw_sum_dpp tmw_sum
#define tmw_readlane_dpp tmw_readlane
#endif
TM_DEV float tmw_sum_dpp(const float *vp) {
  float v = vp[0];
  v = tmw_dpp_add<0x111, 0xf, 0xf>(v);  // row_shr:1
  v = tmw_dpp_add<0x112, 0xf, 0xf>(v);  // row_shr:2
  v = tmw_dpp_add<0x114, 0xf, 0xf>(v);  // row_shr:4
  v = tmw_dpp_add<0x118, 0xf, 0xf>(v);  // row_shr:8
  v = tmw_dpp_add<0x142, 0xf, 0xf>(v);  // row_bcast:15
  v = tmw_dpp_add<0x143, 0xf, 0xf>(v);  // row_bcast:31
  return tm_i2f(__builtin_amdgcn_readlane(tm_f2i(v), 63));  // the builtin is typed int: pass the BITS, not the value
}
TM_DEV float tmw_readlane_dpp(const float *v, int src) { return tm_i2f(__builtin_amdgcn_readlane(tm_f2i(v[0]), src)); }
#endif
// sum over the first WIDTH (16 / 32 / 64, compile time) lanes when the others are known to hold zeros: the two row_bcast steps
// (the expensive ones: v_mov_dpp + add + hazard nops) disappear for WIDTH = 16, one of them for WIDTH = 32
#ifdef TM_HOST_EMU
template <int WIDTH> TM_DEV float tmw_sum_w(const float *v) { return tmw_sum(v); }
#else
template <int WIDTH> TM_DEV float tmw_sum_w(const float *vp) {
  float v = vp[0];
  v = tmw_dpp_add<0x111, 0xf, 0xf>(v);
  v = tmw_dpp_add<0x112, 0xf, 0xf>(v);
  v = tmw_dpp_add<0x114, 0xf, 0xf>(v);
  v = tmw_dpp_add<0x118, 0xf, 0xf>(v);
  if (WIDTH <= 16) return tm_i2f(__builtin_amdgcn_readlane(tm_f2i(v), 15));
  v = tmw_dpp_add<0x142, 0xf, 0xf>(v);
  if (WIDTH <= 32) return tm_i2f(__builtin_amdgcn_readlane(tm_f2i(v), 31));
  v = tmw_dpp_add<0x143, 0xf, 0xf>(v);
  return tm_i2f(__builtin_amdgcn_readlane(tm_f2i(v), 63));
}
#endif
// element `idx` (uniform) of a 128-entry table kept as two lane vectors
#ifdef TM_HOST_EMU
TM_DEV int tmw_table2(const float *a0, const float *a1, int idx) { return tm_f2i(idx < 64 ? a0[idx] : a1[idx - 64]); }
#else
TM_DEV int tmw_table2(const float *a0, const float *a1, int idx) { return __builtin_amdgcn_readlane(tm_f2i(idx < 64 ? a0[0] : a1[0]), idx & 63); }
#endif

// exclusive count of set flags over the lanes of the wave (v_mbcnt of the ballot) and their total
#ifdef TM_HOST_EMU
TM_DEV int tmw_prefix(const int *flag, int *excl) { int s = 0; for (int l = 0; l < 64; l++) { excl[l] = s; s += flag[l] != 0; } return s; }
#else
TM_DEV int tmw_prefix(const int *flag, int *excl) {
  unsigned long long b = __builtin_amdgcn_ballot_w64(flag[0] != 0);
  excl[0] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
  return __builtin_popcountll(b);
}
#endif
#define TMW_ROWMAP(K) ((unsigned char *)(L + (K).l_rowmap))
#define TMW_CCROW(K) ((unsigned char *)(L + (K).l_ccrow))
#define TMW_CONGRP(K) ((unsigned char *)(L + (K).l_con_grpb) + (K).ncon)     /* paw group of a contact slot: behind the ccrow bytes */

// ------------------------------------------------------------------------------------------ state in / out
TM_DEV float tmw_load_state(WCtx &c, const WLayout &K, const float *action) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  // The env's LDS image starts from ZEROS, not from what the previous wave on this CU left there.  The branch-free row products read a few
  // words in front of / behind their vectors and cancel them by a zero FACTOR (tmw_row_runs2: select on the matrix entry only, one instruction
  // per product fewer) — sound while those words are finite, and a wave that ran an env whose state had blown up leaves NaNs behind.  Until
  // round 5 every such word happened to lie in an array the position stage rewrites; with cinert moved onto xipos the tail of region B is not
  // (tests/diagnostics/kernel_ab.py under TMJX_EMU_POISON=nan found it; on the GPU it showed as a flaky twin test under full-scale actions).
  // Once per launch (ten substeps): 45 stores per lane.
  TMW_FOR { for (int i = lane; i < K.lds_floats; i += 64) L[i] = 0.f; }
  TMW_SYNC();
  TMW_FOR {
    for (int i = lane; i < K.nq + K.nv + (K.lean ? 0 : K.nu); i += 64) L[K.l_qpos + i] = WST(m.s_qpos, i);      // qpos | qvel (| act); the warm start stays in global memory
    if (K.lean) {     // the dofs' packed index words: two registers per lane, one LDS word per paw group
      c.tp0[TMW_LI] = m.tpack[lane < K.nv ? lane : 0]; c.tp1[TMW_LI] = m.tpack[lane + 64 < K.nv ? lane + 64 : 64];
      for (int g = lane; g < K.ngroup; g += 64) L[K.l_tgrp + g] = tm_i2f(m.gpack[g]);
    } else {
      for (int i = lane; i < 2 * K.nv; i += 64) {   // index table of the sparse rows (+ the dof's limit row / wrench subset in the top bytes)
        int dof = i >> 1, extra = (i & 1) ? m.dof_wsub[dof] + 1 : m.dof_limrow[dof] + 1;
        L[K.l_tdof + i] = tm_i2f(m.tdof[i] | (extra << 24));
      }
      for (int g = lane; g < K.ngroup; g += 64) ((signed char *)(L + K.l_tgrp))[g] = (signed char)m.grp_lastdof[g];
    }
    for (int cc = lane; cc < K.ncon; cc += 64) { if (!K.lean) L[K.l_con_mu + cc] = m.con_mu[cc]; TMW_CONGRP(K)[cc] = (unsigned char)m.con_grp[cc]; }
    {
      int su = lane / 6;
      bool ok = lane < m.n_wsub * 6;
      c.kmask[TMW_LI] = ok ? m.wsub_cmask[su][0] : 0u;
    }
  }
  TMW_SYNC();
  return WST(m.s_time, 0);
}
// packed per-dof words in LDS: w0 = (Madr + depth) | depth << 16 | (limit row + 1) << 24, w1 = chain_start | (jump + 1) << 8 | ndesc << 16 | (wrench subset + 1) << 24
#define TMW_MEND(w0) ((w0) & 0xffff)              /* address of the LAST entry of the row = Madr + depth */
#define TMW_DEPTH(w0) (((w0) >> 16) & 0xff)
#define TMW_ADR(w0) (((w0) & 0xffff) - TMW_DEPTH(w0)) /* Madr: address of the diagonal entry */
// the top bytes of the two words carry two more per-dof model constants for tmw_jt_force (filled in tmw_load_state): w0 bits 24..31 =
// limit row + 1, w1 bits 24..31 = wrench subset + 1 (0 = none)
#define TMW_LIMROW1(w0) (((w0) >> 24) & 0xff)
#define TMW_WSUB1(w1) (((w1) >> 24) & 0xff)
#define TMW_W0(i) tm_f2i(L[K.l_tdof + 2 * (i)])
#define TMW_W1(i) tm_f2i(L[K.l_tdof + 2 * (i) + 1])
// q-th ancestor of dof i (q = 0: i itself): the chain i, i-1, .., chain_start, then jump, jump-1, .., 0
TM_DEV int tmw_anc(int i, int q, int w1) {
  int r = i - (w1 & 0xff);
  return q <= r ? i - q : ((w1 >> 8) & 0xff) + r - q;
}
// lean layout: the dof's ONE packed word (DModel::tpack) sits in a register of the lane that owns the dof (i = lane: tp0, i = lane + 64: tp1)
// (read through an OPAQUE copy at every use: the words are launch constants, and whatever is decoded from a visible launch constant — depths, row
// addresses, loop masks — is hoisted in front of the substep loop and kept in registers for the whole kernel: 131 -> 158 VGPRs without this)
#ifdef TM_HOST_EMU
TM_DEV int tmw_opaque_v(int x) { return x; }
#else
TM_DEV int tmw_opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
#endif
#define TMW_TP(i) tmw_opaque_v((i) < 64 ? c.tp0[TMW_LI] : c.tp1[TMW_LI])
#define TMW_TP_MEND(p) ((p) & 0x7ff)
#define TMW_TP_DEPTH(p) (((p) >> 11) & 0x3f)
#define TMW_TP_RUN(p) (((p) >> 17) & 0x1f)               /* i - chain_start;  jump + 1 = depth - run */
#define TMW_TP_WSUB1(p) (((p) >> 22) & 0xf)
TM_DEV int tmw_anc_r(int i, int q, int r, int jp1) { return q <= r ? i - q : jp1 + r - q; }
TM_DEV void tmw_store_state(WCtx &c, const WLayout &K, float time) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_FOR {
    for (int i = lane; i < K.nq + K.nv + (K.lean ? 0 : K.nu); i += 64) WST(m.s_qpos, i) = L[K.l_qpos + i];      // (warm start and qfrc_actuator are written where they arise)
    if (lane == 0) WST(m.s_time, 0) = time;
  }
  TMW_SYNC();
}

// contact frame rows n, b of slot cc: 6 words per slot, or (chain layout) the shared normal + 3 words per slot
TM_DEV void tmw_put_con_frame(float *L, const WLayout &K, int cc, const float *fr) {
  if (K.lean) { if (cc == 0) for (int k = 0; k < 3; k++) L[K.l_con_frame + k] = fr[k]; for (int k = 0; k < 3; k++) L[K.l_con_frame + 3 + cc * 3 + k] = fr[3 + k]; }
  else for (int k = 0; k < 6; k++) L[K.l_con_frame + cc * 6 + k] = fr[k];
}
TM_DEV void tmw_get_con_frame(const float *L, const WLayout &K, int cc, float *fr) {
  if (K.lean) { for (int k = 0; k < 3; k++) { fr[k] = L[K.l_con_frame + k]; fr[3 + k] = L[K.l_con_frame + 3 + cc * 3 + k]; } }
  else for (int k = 0; k < 6; k++) fr[k] = L[K.l_con_frame + cc * 6 + k];
}

// ------------------------------------------------------------------------------------------ fwd_position
// kinematics by pointer jumping; com; collision; cdof; cinert.  `emit`: also write xpos / torso xmat to global.
TM_DEV void tmw_position(WCtx &c, const WLayout &K, bool emit) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  // (1) local transform of every body relative to its parent (absolute for the free-joint body).  Everything the lane needs of the model for
  // its bodies comes from ONE flat record per body (DModel::body_kin), both bodies' records loaded up front: the chain body -> joint ->
  // joint fields -> qpos0[qposadr] was three dependent model reads per body, each about a thousand cycles next to eleven other waves
  // Head-4 body tree (K.lean; model_host.h: rodent_chains_match checks it): body 0 = world (identity), 1 = a static child of the world, 2 = the
  // free-joint root (absolute by its qpos), 3 = welded to it; the other nbody - 4 <= 64 bodies take ONE lane each — a second body slot for
  // bodies 64 .. 67 cost every instruction of steps (1) and (2) a second time for four live lanes.  The four head bodies are finished by lanes
  // 0 .. 3 right here: their entries never change during the rounds, except body 3's, which is its local frame in round 0 and root o local
  // from round 1 on — the same numbers the two-slot form produced (bit-identical: tests/diagnostics/kernel_ab.py).
  const int HB = K.lean ? 4 : 0;
  TMW_FOR {
    float rec[2][16];
#pragma unroll
    for (int slot = 0; slot < (K.lean ? 1 : 2); slot++) {
      const int b = lane + HB + 64 * slot < K.nbody ? lane + HB + 64 * slot : K.nbody - 1;
#pragma unroll
      for (int k = 0; k < 16; k++) rec[slot][k] = m.body_kin[b][k];
    }
#pragma unroll
    for (int slot = 0; slot < (K.lean ? 1 : 2); slot++) {
      const int b = lane + HB + 64 * slot;
      if (b >= K.nbody) continue;
      const float *rc = rec[slot];
      float t[3] = {rc[0], rc[1], rc[2]};
      float q[4] = {rc[3], rc[4], rc[5], rc[6]};
      for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
        int j = m.body_jntadr[b] + jj, qa, jtype;
        float jpos[3], jaxis[3], q0;
        if (jj == 0) {
          jtype = tm_f2i(rc[7]); qa = tm_f2i(rc[8]); q0 = rc[15];
          for (int k = 0; k < 3; k++) { jpos[k] = rc[9 + k]; jaxis[k] = rc[12 + k]; }
        } else {          // (a body with several joints: the others through the tables)
          jtype = m.jnt_type[j]; qa = m.jnt_qposadr[j]; q0 = m.qpos0[qa];
          for (int k = 0; k < 3; k++) { jpos[k] = m.jnt_pos[j][k]; jaxis[k] = m.jnt_axis[j][k]; }
        }
        if (!K.lean && jtype == 0) {
          for (int k = 0; k < 3; k++) t[k] = L[K.l_qpos + qa + k];
          for (int k = 0; k < 4; k++) q[k] = L[K.l_qpos + qa + 3 + k];
          tm_normalize4(q);
          for (int k = 0; k < 4; k++) L[K.l_qpos + qa + 3 + k] = q[k];
          for (int k = 0; k < 3; k++) { L[K.l_jl_anchor + j * 3 + k] = t[k]; L[K.l_jl_axis + j * 3 + k] = (k == 2) ? 1.f : 0.f; }
        } else {
          float r[3], an[3], ax[3], ql[4], q2[4];
          tm_rotate(r, jpos, q);
          for (int k = 0; k < 3; k++) an[k] = t[k] + r[k];
          tm_rotate(ax, jaxis, q);
          float ang = (L[K.l_qpos + qa] - q0) * 0.5f, sn, cs;
          sincosf(ang, &sn, &cs);      // one shared range reduction
          ql[0] = cs; ql[1] = jaxis[0] * sn; ql[2] = jaxis[1] * sn; ql[3] = jaxis[2] * sn;
          tm_quat_mul(q2, q, ql);
          for (int k = 0; k < 4; k++) q[k] = q2[k];
          tm_rotate(r, jpos, q);
          for (int k = 0; k < 3; k++) { t[k] = an[k] - r[k]; L[K.l_jl_anchor + j * 3 + k] = an[k]; L[K.l_jl_axis + j * 3 + k] = ax[k]; }
        }
      }
      float *o = L + K.l_scanA + b * 8;
      o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3]; o[7] = tm_i2f(m.scan_parent[b]);
    }
    if (K.lean && lane < 4) {
      const int b = lane, jr = m.body_jntadr[2], qr = tm_f2i(m.body_kin[2][8]);
      float t[3], q[4], tr[3], qq[4];       // the root's frame, by every one of the four lanes (lane 3 composes with it)
      for (int k = 0; k < 3; k++) tr[k] = L[K.l_qpos + qr + k];
      for (int k = 0; k < 4; k++) qq[k] = L[K.l_qpos + qr + 3 + k];
      tm_normalize4(qq);
      for (int k = 0; k < 3; k++) t[k] = m.body_kin[b][k];
      for (int k = 0; k < 4; k++) q[k] = m.body_kin[b][3 + k];
      if (b == 2) {
        for (int k = 0; k < 3; k++) t[k] = tr[k];
        for (int k = 0; k < 4; k++) q[k] = qq[k];
        for (int k = 0; k < 3; k++) { L[K.l_jl_anchor + jr * 3 + k] = tr[k]; L[K.l_jl_axis + jr * 3 + k] = (k == 2) ? 1.f : 0.f; }
      }
      float *o = L + K.l_scanA + b * 8;
      o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3]; o[7] = tm_i2f(b == 3 ? 2 : -1);
      if (b == 3) {
        float rt[3], q2[4];
        tm_rotate(rt, t, qq);
        for (int k = 0; k < 3; k++) t[k] = tr[k] + rt[k];
        tm_quat_mul(q2, qq, q);
        for (int k = 0; k < 4; k++) q[k] = q2[k];
      }
      o = L + K.l_scanB + b * 8;
      o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3]; o[7] = tm_i2f(-1);
    }
  }
  TMW_SYNC();
  // (the root's normalised quaternion back into qpos, as the free-joint branch does: behind the barrier — every head lane has read the raw one)
  if (K.lean) { TMW_FOR { if (lane < 4) L[K.l_qpos + tm_f2i(m.body_kin[2][8]) + 3 + lane] = L[K.l_scanA + 2 * 8 + 3 + lane]; } }
  TMW_TICK2(27);
  // (2) pointer jumping: T_b <- T_anc(b) o T_b ; anc(b) <- anc(anc(b)).  Even round count => result in scanA.
  int R = K.nround_body + (K.nround_body & 1);
  for (int r = 0; r < R; r++) {
    const float *cur = L + ((r & 1) ? K.l_scanB : K.l_scanA);
    float *nxt = L + ((r & 1) ? K.l_scanA : K.l_scanB);
    TMW_FOR {
      for (int b = lane + HB; b < K.nbody; b += 64) {
        const float *s = cur + b * 8;
        float t[3] = {s[0], s[1], s[2]}, q[4] = {s[3], s[4], s[5], s[6]};
        int a = tm_f2i(s[7]);
        if (a >= 0) {
          const float *sa = cur + a * 8;
          float qa[4] = {sa[3], sa[4], sa[5], sa[6]}, rt[3], q2[4];
          tm_rotate(rt, t, qa);
          for (int k = 0; k < 3; k++) t[k] = sa[k] + rt[k];
          tm_quat_mul(q2, qa, q);
          for (int k = 0; k < 4; k++) q[k] = q2[k];
          a = tm_f2i(sa[7]);
        }
        float *o = nxt + b * 8;
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3]; o[7] = tm_i2f(a);
      }
      if (K.lean && r == 1 && lane < 8) nxt[3 * 8 + lane] = cur[3 * 8 + lane];      // body 3's finished frame into scanA as well (the other head entries are in both buffers)
    }
    TMW_SYNC();
  }
  TMW_TICK2(28);
  // (3) inertial frame origins and the tree's centre of mass
  TMW_REG(float, s0); TMW_REG(float, s1); TMW_REG(float, s2);
  TMW_FOR {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = lane; b < K.nbody; b += 64) {
      const float *s = L + K.l_scanA + b * 8;
      float r[3];
      { TMW_LOCAL(ip_, 3, m.body_ipos[b]); tm_rotate(r, ip_, s + 3); }
      float x0 = s[0] + r[0], x1 = s[1] + r[1], x2 = s[2] + r[2];
      L[K.l_xipos + b * 3] = x0; L[K.l_xipos + b * 3 + 1] = x1; L[K.l_xipos + b * 3 + 2] = x2;
      if (m.body_moving[b]) { float mb = m.body_mass[b]; a0 += x0 * mb; a1 += x1 * mb; a2 += x2 * mb; }
      if (emit) {
        for (int k = 0; k < 3; k++) WST(m.s_xpos, b * 3 + k) = s[k];
        if (b == m.torso_idx) { float X[9]; tm_quat_to_mat(X, s + 3); for (int k = 0; k < 9; k++) WST(m.s_xmat_torso, k) = X[k]; }
      }
    }
    s0[TMW_LI] = a0; s1[TMW_LI] = a1; s2[TMW_LI] = a2;
  }
  float com[3] = {tmw_sum(s0) / m.total_mass, tmw_sum(s1) / m.total_mass, tmw_sum(s2) / m.total_mass};
  TMW_FOR { if (lane < 3) L[K.l_com + lane] = com[lane]; }
  TMW_TICK2(29);
  // (4) collision: one lane per contact slot (plane vs paw capsule end / ellipsoid)
  TMW_FOR {
    for (int cc = lane; cc < K.ncon; cc += 64) {
      int b1 = m.con_body1[cc], b2 = m.con_body2[cc];
      const float *s1p = L + K.l_scanA + b1 * 8, *s2p = L + K.l_scanA + b2 * 8;
      float t[3], pp[3], pq[4], pm[9], gp[3], gq[4], gm[9];
      { TMW_LOCAL(gp_, 3, m.con_g1_pos[cc]); tm_rotate(t, gp_, s1p + 3); }
      for (int k = 0; k < 3; k++) pp[k] = s1p[k] + t[k];
      { TMW_LOCAL(gq_, 4, m.con_g1_quat[cc]); tm_quat_mul(pq, s1p + 3, gq_); } tm_quat_to_mat(pm, pq);
      { TMW_LOCAL(gp_, 3, m.con_g2_pos[cc]); tm_rotate(t, gp_, s2p + 3); }
      for (int k = 0; k < 3; k++) gp[k] = s2p[k] + t[k];
      { TMW_LOCAL(gq_, 4, m.con_g2_quat[cc]); tm_quat_mul(gq, s2p + 3, gq_); } tm_quat_to_mat(gm, gq);
      float nrm[3] = {pm[2], pm[5], pm[8]}, fr[9], pos[3], dist;
      TMW_LOCAL(size, 3, m.con_g2_size[cc]);
      if (m.con_type[cc] == 3) {
        float axis[3] = {gm[2], gm[5], gm[8]}, bb[3], na = tm_dot3(nrm, axis), cr[3];
        for (int k = 0; k < 3; k++) bb[k] = axis[k] - nrm[k] * na;
        float bn = tm_normalize3(bb);
        if (bn < 0.5f) { bool yy = (-0.5f < nrm[1]) && (nrm[1] < 0.5f); bb[0] = 0.f; bb[1] = yy ? 1.f : 0.f; bb[2] = yy ? 0.f : 1.f; }
        tm_cross(cr, nrm, bb);
        for (int k = 0; k < 3; k++) { fr[k] = nrm[k]; fr[3 + k] = bb[k]; fr[6 + k] = cr[k]; }
        float sg = m.con_sub[cc] == 0 ? 1.f : -1.f, end[3];
        for (int k = 0; k < 3; k++) end[k] = gp[k] + sg * (axis[k] * size[1]);
        tm_plane_sphere(nrm, pp, end, size[0], dist, pos);
      } else if (m.con_type[cc] == 4) {
        float loc[3], sup[3], w[3], df[3];
        for (int k = 0; k < 3; k++) loc[k] = gm[k] * nrm[0] + gm[3 + k] * nrm[1] + gm[6 + k] * nrm[2];
        for (int k = 0; k < 3; k++) sup[k] = loc[k] * size[k];
        tm_normalize3(sup);
        for (int k = 0; k < 3; k++) sup[k] = -sup[k] * size[k];
        for (int k = 0; k < 3; k++) w[k] = gm[k * 3] * sup[0] + gm[k * 3 + 1] * sup[1] + gm[k * 3 + 2] * sup[2];
        for (int k = 0; k < 3; k++) { pos[k] = gp[k] + w[k]; df[k] = pos[k] - pp[k]; }
        dist = tm_dot3(nrm, df);
        for (int k = 0; k < 3; k++) pos[k] = pos[k] - nrm[k] * dist * 0.5f;
        tm_make_frame(nrm, fr);
      } else {
        tm_plane_sphere(nrm, pp, gp, size[0], dist, pos);
        tm_make_frame(nrm, fr);
      }
      L[K.l_con_dist + cc] = dist;
      for (int k = 0; k < 3; k++) L[K.l_con_off + cc * 3 + k] = pos[k] - com[k];
      tmw_put_con_frame(L, K, cc, fr);   // rows n and b; the third row is n x b
    }
  }
  TMW_TICK2(30);
  // (5) cdof: one lane per dof (joint anchors / axes are stored in the parent frame); the dof's joint type / body / parent body / index inside
  // the joint from its flat record (DModel::dof_kin: one model read instead of dof -> joint -> body -> parent)
  TMW_FOR {
    int dk[2][4];
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
      const int i = lane + 64 * slot < K.nv ? lane + 64 * slot : K.nv - 1;
#pragma unroll
      for (int k = 0; k < 4; k++) dk[slot][k] = m.dof_kin[i][k];
    }
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
      const int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      const int jtype = dk[slot][0], b = dk[slot][1], p = dk[slot][2], k = dk[slot][3], j = m.dof_jntid[i];
      float cd[6];
      if (jtype == 0) {
        const float *s = L + K.l_scanA + b * 8;
        if (k < 3) { for (int r = 0; r < 6; r++) cd[r] = (r == 3 + k) ? 1.f : 0.f; }
        else {
          float X[9], off[3] = {com[0] - s[0], com[1] - s[1], com[2] - s[2]};
          tm_quat_to_mat(X, s + 3);
          float ax[3] = {X[k - 3], X[3 + k - 3], X[6 + k - 3]}, cr[3];
          tm_cross(cr, ax, off);
          for (int r = 0; r < 3; r++) { cd[r] = ax[r]; cd[3 + r] = cr[r]; }
        }
      } else {
        const float *sp = L + K.l_scanA + p * 8;
        float an[3], ax[3], off[3], cr[3];
        tm_rotate(an, L + K.l_jl_anchor + j * 3, sp + 3);
        tm_rotate(ax, L + K.l_jl_axis + j * 3, sp + 3);
        for (int r = 0; r < 3; r++) off[r] = com[r] - (sp[r] + an[r]);
        tm_cross(cr, ax, off);
        for (int r = 0; r < 3; r++) { cd[r] = ax[r]; cd[3 + r] = cr[r]; }
      }
      for (int r = 0; r < 6; r++) L[K.l_cdof + i * 6 + r] = cd[r];
    }
  }
  TMW_SYNC();
  TMW_TICK2(31);
  // (6) cinert (overwrites the dead scanB / joint-frame scratch — and, in the lean layout, xipos itself: every lane takes the inertial-frame
  // origins of its two bodies into registers first; round 5, wave_layout.h: l_cinert = l_xipos)
  TMW_REG(float, xo0); TMW_REG(float, xo1); TMW_REG(float, xo2); TMW_REG(float, xo3); TMW_REG(float, xo4); TMW_REG(float, xo5);
  TMW_FOR {
    const int b0 = lane < K.nbody ? lane : 0, b1 = lane + 64 < K.nbody ? lane + 64 : 0;
    xo0[TMW_LI] = L[K.l_xipos + b0 * 3]; xo1[TMW_LI] = L[K.l_xipos + b0 * 3 + 1]; xo2[TMW_LI] = L[K.l_xipos + b0 * 3 + 2];
    xo3[TMW_LI] = L[K.l_xipos + b1 * 3]; xo4[TMW_LI] = L[K.l_xipos + b1 * 3 + 1]; xo5[TMW_LI] = L[K.l_xipos + b1 * 3 + 2];
  }
  TMW_SYNC();
  TMW_FOR {
    for (int b = lane; b < K.nbody; b += 64) {
      float ci[10];
      // (the model reads of a body are issued before the `moving` test: behind it they were a second dependent level)
      const int moving = m.body_moving[b];
      TMW_LOCAL(iq_, 4, m.body_iquat[b]);
      TMW_LOCAL(in, 3, m.body_inertia[b]);
      const float mass = m.body_mass[b];
      if (!moving) { for (int k = 0; k < 10; k++) ci[k] = 0.f; }
      else {
        const float *s = L + K.l_scanA + b * 8;
        float q[4], X[9], off[3];
        tm_quat_mul(q, s + 3, iq_);
        tm_quat_to_mat(X, q);
        off[0] = (b < 64 ? xo0[TMW_LI] : xo3[TMW_LI]) - com[0]; off[1] = (b < 64 ? xo1[TMW_LI] : xo4[TMW_LI]) - com[1]; off[2] = (b < 64 ? xo2[TMW_LI] : xo5[TMW_LI]) - com[2];
        float oo = tm_dot3(off, off);
        ci[0] = X[0] * in[0] * X[0] + X[1] * in[1] * X[1] + X[2] * in[2] * X[2] + (oo - off[0] * off[0]) * mass;
        ci[1] = X[3] * in[0] * X[3] + X[4] * in[1] * X[4] + X[5] * in[2] * X[5] + (oo - off[1] * off[1]) * mass;
        ci[2] = X[6] * in[0] * X[6] + X[7] * in[1] * X[7] + X[8] * in[2] * X[8] + (oo - off[2] * off[2]) * mass;
        ci[3] = X[0] * in[0] * X[3] + X[1] * in[1] * X[4] + X[2] * in[2] * X[5] - off[0] * off[1] * mass;
        ci[4] = X[0] * in[0] * X[6] + X[1] * in[1] * X[7] + X[2] * in[2] * X[8] - off[0] * off[2] * mass;
        ci[5] = X[3] * in[0] * X[6] + X[4] * in[1] * X[7] + X[5] * in[2] * X[8] - off[1] * off[2] * mass;
        ci[6] = off[0] * mass; ci[7] = off[1] * mass; ci[8] = off[2] * mass; ci[9] = mass;
      }
      // cinert rows never overlap scanA (the matrix region in the lean layout); xipos, which they may overlay, is in registers since the barrier above
      for (int k = 0; k < 10; k++) L[K.l_cinert + b * 10 + k] = ci[k];
    }
  }
  TMW_SYNC();
}

// Inclusive prefix sums along the dof tree for the CHAIN layout, in place on scan entries of TMW_DS floats at `base` (6-vector per
// dof): one lane per (chain, component) walks its chain serially — the trunk first, then every leaf chain starting from the trunk's
// finished value at its attachment point.  36 dependent-free load / add / store steps instead of 6 pointer-jumping rounds over all
// dofs x two dof slots with a barrier each (the loads do not depend on the running sum, the stores are fire-and-forget).
TM_DEV void tmw_chain_scan(WCtx &c, const WLayout &K, int base) {
  float *L = c.L; TMW_LANE_DECL
  // (round 6: the loads of a block of steps are issued TOGETHER, then the dependent adds and the stores — one load / wait / add / store per
  // trip made every step a whole LDS round trip next to eleven other waves: ~ 200 cycles x 36 steps x two scans per substep)
  TMW_FOR {
    if (lane < 6) {
      float *p = L + base + lane, v[TMW_RODENT_TRUNK], acc = 0.f;
#pragma unroll
      for (int t = 0; t < TMW_RODENT_TRUNK; t++) v[t] = p[t * TMW_DS];
#pragma unroll
      for (int t = 0; t < TMW_RODENT_TRUNK; t++) { acc += v[t]; p[t * TMW_DS] = acc; }
    }
  }
  TMW_SYNC();
  TMW_FOR {
    if (lane < TMW_RODENT_NCHAIN * 6) {
      const int ci = lane / 6, k = lane - ci * 6;
      int first = 0, n = 0, d0 = 1, idx = 0;
#define TMW_X(f, nn, dd) if (ci == idx) { first = f; n = nn; d0 = dd; } idx++;
      TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
      float acc = L[base + (d0 - 1) * TMW_DS + k];
      float *p = L + base + first * TMW_DS + k;
      constexpr int MAXN = tmw_chain_maxrun(TMW_RODENT_TRUNK, 73) + 1, UB = 8;        // longest leaf chain (TMW_RODENT_DIMS: nv = 73)
      float keep = 0.f;
#pragma unroll 1
      for (int t0 = 0; t0 < MAXN; t0 += UB) {       // branch-free: a step t >= n re-reads the chain's LAST entry and stores the finished last sum there again
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) v[u] = p[(t0 + u < n ? t0 + u : n - 1) * TMW_DS];
#pragma unroll
        for (int u = 0; u < UB; u++) { acc += v[u]; keep = t0 + u < n ? acc : keep; p[(t0 + u < n ? t0 + u : n - 1) * TMW_DS] = keep; }
      }
    }
  }
  TMW_SYNC();
}

// ------------------------------------------------------------------------------------------ fwd_velocity + smooth forces
// com_vel / rne prefixes by pointer jumping over dofs, body forces, up-sweep (crb, cfrc), M, qfrc_smooth, act_dot
TM_DEV void tmw_velocity_inertia(WCtx &c, const WLayout &K) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  int R = K.nround_dof + (K.nround_dof & 1);
  // inclusive prefix P_i = sum over ancestors-or-self of cdof * qvel
  TMW_FOR {      // (both dof slots' loads first: TMW_I1)
    const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
    const float qv0 = L[K.l_qvel + lane], qv1 = L[K.l_qvel + i1];
    float cd0[6], cd1[6];
    for (int k = 0; k < 6; k++) { cd0[k] = L[K.l_cdof + lane * 6 + k]; cd1[k] = L[K.l_cdof + i1 * 6 + k]; }
    float *o0 = L + K.l_dscanA + lane * TMW_DS, *o1 = L + K.l_dscanA + i1 * TMW_DS;
    for (int k = 0; k < 6; k++) o0[k] = cd0[k] * qv0;
    if (!K.chains) o0[6] = tm_i2f(m.dof_parentid[lane]);
    if (ok1) { for (int k = 0; k < 6; k++) o1[k] = cd1[k] * qv1; if (!K.chains) o1[6] = tm_i2f(m.dof_parentid[i1]); }
  }
  TMW_SYNC();
  if (K.chains) tmw_chain_scan(c, K, K.l_dscanA);
  else
  for (int r = 0; r < R; r++) {
    const float *cur = L + ((r & 1) ? K.l_dscanB : K.l_dscanA);
    float *nxt = L + ((r & 1) ? K.l_dscanA : K.l_dscanB);
    TMW_FOR {
      for (int i = lane; i < K.nv; i += 64) {
        const float *s = cur + i * TMW_DS;
        float v[6] = {s[0], s[1], s[2], s[3], s[4], s[5]};
        int a = tm_f2i(s[6]);
        if (a >= 0) { const float *sa = cur + a * TMW_DS; for (int k = 0; k < 6; k++) v[k] += sa[k]; a = tm_f2i(sa[6]); }
        float *o = nxt + i * TMW_DS;
        for (int k = 0; k < 6; k++) o[k] = v[k];
        o[6] = tm_i2f(a);
      }
    }
    TMW_SYNC();
  }
  TMW_TICK2(20);
  // cdof_dot (kept in registers, two dof slots per lane) and body velocities
  float dd0[TMW_NL][6], dd1[TMW_NL][6], cv0[TMW_NL][6], cv1[TMW_NL][6];
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int i = lane + 64 * slot;
      float *dd = slot ? dd1[TMW_LI] : dd0[TMW_LI];
      for (int k = 0; k < 6; k++) dd[k] = 0.f;
      if (i < K.nv && !m.dof_freetrans[i]) {
        int vp = m.dof_vpar[i];
        float E[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cd[6];
        if (vp >= 0) for (int k = 0; k < 6; k++) E[k] = L[K.l_dscanA + vp * TMW_DS + k];
        for (int k = 0; k < 6; k++) cd[k] = L[K.l_cdof + i * 6 + k];
        tm_motion_cross(dd, E, cd);
      }
      int b = lane + 64 * slot;
      float *cv = slot ? cv1[TMW_LI] : cv0[TMW_LI];
      for (int k = 0; k < 6; k++) cv[k] = 0.f;
      if (b < K.nbody) { int ld = m.body_lastdof[b]; if (ld >= 0) for (int k = 0; k < 6; k++) cv[k] = L[K.l_dscanA + ld * TMW_DS + k]; }
    }
  }
  TMW_SYNC();
  // inclusive prefix Q_i = sum of cdof_dot * qvel
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      const float *dd = slot ? dd1[TMW_LI] : dd0[TMW_LI];
      float qv = L[K.l_qvel + i];
      float *o = L + K.l_dscanA + i * TMW_DS;
      for (int k = 0; k < 6; k++) o[k] = dd[k] * qv;
      if (!K.chains) o[6] = tm_i2f(m.dof_parentid[i]);
    }
  }
  TMW_SYNC();
  if (K.chains) tmw_chain_scan(c, K, K.l_dscanA);
  else
  for (int r = 0; r < R; r++) {
    const float *cur = L + ((r & 1) ? K.l_dscanB : K.l_dscanA);
    float *nxt = L + ((r & 1) ? K.l_dscanA : K.l_dscanB);
    TMW_FOR {
      for (int i = lane; i < K.nv; i += 64) {
        const float *s = cur + i * TMW_DS;
        float v[6] = {s[0], s[1], s[2], s[3], s[4], s[5]};
        int a = tm_f2i(s[6]);
        if (a >= 0) { const float *sa = cur + a * TMW_DS; for (int k = 0; k < 6; k++) v[k] += sa[k]; a = tm_f2i(sa[6]); }
        float *o = nxt + i * TMW_DS;
        for (int k = 0; k < 6; k++) o[k] = v[k];
        o[6] = tm_i2f(a);
      }
    }
    TMW_SYNC();
  }
  TMW_TICK2(21);
  // body forces: cfrc_b = I_b cacc_b + cvel_b x* (I_b cvel_b)
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int b = lane + 64 * slot;
      if (b >= K.nbody) continue;
      const float *cv = slot ? cv1[TMW_LI] : cv0[TMW_LI];
      float ca[6] = {0.f, 0.f, 0.f, -m.gravity[0], -m.gravity[1], -m.gravity[2]}, I[10], f1[6], t[6], f2[6];
      int ld = m.body_lastdof[b];
      if (ld >= 0) for (int k = 0; k < 6; k++) ca[k] += L[K.l_dscanA + ld * TMW_DS + k];
      for (int k = 0; k < 10; k++) I[k] = L[K.l_cinert + b * 10 + k];
      tm_inert_mul(f1, I, ca);
      tm_inert_mul(t, I, cv);
      tm_motion_cross_force(f2, cv, t);
      for (int k = 0; k < 6; k++) L[K.l_cfrc + b * 6 + k] = (b == 0) ? 0.f : f1[k] + f2[k];
    }
  }
  TMW_SYNC();
  TMW_TICK2(22);
  // composite inertia / accumulated body force of every subtree, in place (cinert[b], cfrc[b] <- sum over the subtree of b).
  // Bodies are numbered depth-first; a RUN is a maximal chain b, b+1, .. with parent[b+1] == b.  (1) One branch-free reverse
  // sweep, lane & 15 = component (10 + 6), gives the suffix sums of every run: the running sum is kept or reset by the
  // "has a first child" flag, values are fetched 8 bodies ahead.  (2) The few branch bodies then add the finished sums of
  // their other children to their own run (DModel::fix_*, descending).  All 64 lanes run the sweep, lanes 16.. as copies,
  // and store the same words (no exec-mask branch per store).  This replaces per-dof range sums in which the six root-dof
  // lanes each added up all 67 bodies x 16 components.
  TMW_REG(float, ns0); TMW_REG(float, ns1);
  TMW_FOR { ns0[TMW_LI] = tm_i2f(lane < K.nbody ? m.body_nsub[lane] : 1); ns1[TMW_LI] = tm_i2f(lane + 64 < K.nbody ? m.body_nsub[lane + 64] : 1); }
  TMW_FOR {
    const int comp = lane & 15;
    const int base = comp < 10 ? K.l_cinert + comp : K.l_cfrc + comp - 10, stride = comp < 10 ? 10 : 6;
    float run = 0.f;
    for (int b0 = K.nbody - 1; b0 >= 1; b0 -= 8) {
      float vb[8]; int nb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { int b = b0 - u > 0 ? b0 - u : 0; vb[u] = L[base + b * stride]; nb[u] = tmw_table2(ns0, ns1, b); }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int b = b0 - u > 0 ? b0 - u : 0;       // b == 0 (the world body) only pads the last block: its sum is never used
        run = vb[u] + (nb[u] > 1 ? run : 0.f);
        if (TMW_DUP_LANES_OK || lane < 16) L[base + b * stride] = run;
      }
    }
  }
  TMW_SYNC();
  for (int f = 0; f < m.n_fix; f++) {
    int p = m.fix_p[f], r0 = m.fix_r0[f];
    TMW_FOR {
      const int comp = lane & 15;
      const int base = comp < 10 ? K.l_cinert + comp : K.l_cfrc + comp - 10, stride = comp < 10 ? 10 : 6;
      float delta = 0.f;
      for (int e = m.fix_adr[f]; e < m.fix_adr[f + 1]; e++) delta += L[base + m.fix_child[e] * stride];
      for (int b = r0 + (lane >> 4); b <= p; b += 4) L[base + b * stride] += delta;
    }
    TMW_SYNC();
  }
  // lean layout: the activation state lives in the env's global record, each value written and re-read by its OWN lane (tmw_euler) — a dof's
  // lane needs other lanes' values, and global memory gives no ordering between the lanes of a wave: stage it through the paw-velocity
  // buffer, which only the J products of the solver stage use (nu <= 6 ngroup: WLayout::lean)
  if (K.lean) { TMW_FOR { for (int a = lane; a < K.nu; a += 64) L[K.l_sv + a] = WST(m.s_act, a); } }
  TMW_SYNC();
  TMW_TICK2(23);
  // M rows, bias, passive, actuation -> qfrc_smooth; act_dot.  The bias forces first, into registers: in the lean layout the accumulated body
  // forces sit INSIDE the matrix region (wave_layout.h), which the rows of M are about to overwrite
  TMW_REG(float, bias0); TMW_REG(float, bias1);
  TMW_FOR {
    const int i1 = TMW_I1(K), b0 = m.dof_bodyid[lane], b1 = m.dof_bodyid[i1];      // (one level for both slots' model reads, then both slots' LDS reads)
    float cd0[6], cd1[6], cf0[6], cf1[6], s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < 6; k++) { cd0[k] = L[K.l_cdof + lane * 6 + k]; cd1[k] = L[K.l_cdof + i1 * 6 + k]; cf0[k] = L[K.l_cfrc + b0 * 6 + k]; cf1[k] = L[K.l_cfrc + b1 * 6 + k]; }
    for (int k = 0; k < 6; k++) s0 += cd0[k] * cf0[k];
    for (int k = 0; k < 6; k++) s1 += cd1[k] * cf1[k];
    bias0[TMW_LI] = s0; if (TMW_OK1(K)) bias1[TMW_LI] = s1;
  }
  TMW_SYNC();
  TMW_FOR {
    float dd[2][8];          // the dofs' flat records (DModel::dof_dyn), both slots up front
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
      const int i = lane + 64 * slot < K.nv ? lane + 64 * slot : K.nv - 1;
#pragma unroll
      for (int k = 0; k < 8; k++) dd[slot][k] = m.dof_dyn[i][k];
    }
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
      const int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      const float *rc = dd[slot];
      const int b = tm_f2i(rc[0]);
      float I[10], cd[6], buf[6];
      for (int k = 0; k < 10; k++) I[k] = L[K.l_cinert + b * 10 + k];
      for (int k = 0; k < 6; k++) cd[k] = L[K.l_cdof + i * 6 + k];
      tm_inert_mul(buf, I, cd);
      int adr, d, run, jp1;
      if (K.lean) { const int tp = TMW_TP(i); d = TMW_TP_DEPTH(tp); adr = TMW_TP_MEND(tp) - d; run = TMW_TP_RUN(tp); jp1 = d - run; }
      else { const int w0 = TMW_W0(i), w1 = TMW_W1(i); adr = TMW_ADR(w0); d = TMW_DEPTH(w0); run = i - (w1 & 0xff); jp1 = (w1 >> 8) & 0xff; }
#pragma unroll 4
      for (int k = 0; k <= d; k++) {
        const float *cj = L + K.l_cdof + tmw_anc_r(i, k, run, jp1) * 6;
        float s = buf[0] * cj[0] + buf[1] * cj[1] + buf[2] * cj[2] + buf[3] * cj[3] + buf[4] * cj[4] + buf[5] * cj[5];
        if (k == 0) s += rc[1];          // armature
        L[K.l_M + adr + k] = s;
      }
      const float bias = i < 64 ? bias0[TMW_LI] : bias1[TMW_LI];
      float fa = 0.f;
      for (int e = tm_f2i(rc[6]); e < tm_f2i(rc[7]); e++) { int u = m.dof_act_id[e]; fa += m.dof_act_coef[e] * (m.dof_act_gain[e] * L[(K.lean ? K.l_sv : K.l_act) + u]); }
      WST(m.s_qfrc_actuator, i) = fa;
      float f = -rc[2] * L[K.l_qvel + i] - bias + fa;
      if (rc[3] != 0.f) f += -rc[3] * (L[K.l_qpos + tm_f2i(rc[4])] - rc[5]);
      TMW_QFS_SET(i, f);
    }
    // (act_dot = (clamp(ctrl) - act) / tau is formed in tmw_euler, where act is advanced: act does not change in between)
  }
  TMW_SYNC();
}

// ------------------------------------------------------------------------------------------ factorisation, L^-1, solves
// LD <- L^T D L of (M + hdamp diag(damping)); pivots leaf -> root; lane q owns column q of pivot row k
// LD <- L^T D L of (M + hdamp diag(damping)), LEFT-LOOKING: rows are finished leaf -> root; row i gathers, in registers,
// the rank-1 contributions of all its (already final) descendant rows k:  row_i -= (M'(k,i) / D_k) * M'(k, anc(i)).
// Lane q owns column q of row i; lane t pre-computes multiplier and row address of descendant t, the inner loop fetches
// them with scalar readlanes: per (row, descendant) pair one LDS read and one FMA per lane, no LDS write, no masks.
// `rhs` >= 0: an LDS vector eliminated alongside (x <- L^-T x, the leaf -> root sweep of mj_solveLD), so that one more
// root -> leaf pass (tmw_subst_down) completes (M + hD)^-1 rhs without inverting L.
TM_DEV void tmw_factor(WCtx &c, const WLayout &K, float hdamp, int rhs = -1) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_FOR {
    for (int i = lane; i < K.nnz; i += 64) L[K.l_LD + i] = L[K.l_M + i];
  }
  TMW_SYNC();
  if (hdamp != 0.f) {
    TMW_FOR { for (int i = lane; i < K.nv; i += 64) L[K.l_LD + m.dof_Madr[i]] += hdamp * m.dof_damping[i]; }
    TMW_SYNC();
  }
  TMW_REG(float, a0); TMW_REG(float, a1); TMW_REG(float, b0); TMW_REG(float, b1); TMW_REG(float, acc); TMW_REG(float, pr);
  for (int i = K.nv - 1; i >= 0; i--) {
    int w0 = TMW_W0(i), adr = TMW_ADR(w0), d = TMW_DEPTH(w0), nd = (TMW_W1(i) >> 16) & 0xff;
    TMW_FOR {
      float rsum = 0.f;
#pragma unroll
      for (int slot = 0; slot < 2; slot++) {
        int t = lane + 64 * slot;
        float a = 0.f; int base = K.l_LD;
        if (t < nd) {
          // farthest descendant first: the small contributions of the leaves are summed before the large ones of the
          // near descendants (same order as the right-looking elimination; measurably more accurate in fp32)
          int kk = i + nd - t, wk = TMW_W0(kk);
          base = K.l_LD + TMW_MEND(wk) - d;      // M'(k, i) sits at Madr_k + depth_k - depth_i
          a = L[base] * L[K.l_Dinv + kk];                       // = L(k, i)
          if (rhs >= 0) rsum += a * L[rhs + kk];
        }
        (slot ? a1 : a0)[TMW_LI] = a; (slot ? b1 : b0)[TMW_LI] = tm_i2f(base);
      }
      acc[TMW_LI] = (lane <= d) ? L[K.l_LD + adr + lane] : 0.f;
      pr[TMW_LI] = rsum;
    }
    TMW_TICK2(13);
    if (rhs >= 0 && nd > 0) { float s = tmw_sum(pr); TMW_FOR { if (lane == 0) L[rhs + i] -= s; } }
    for (int slot = 0; slot < 2; slot++) {       // descendants 0..63 live in (a0, b0), 64.. in (a1, b1): no per-item select
      const float *ar = slot ? a1 : a0, *br = slot ? b1 : b0;
      int cnt = nd - 64 * slot; if (cnt > 64) cnt = 64;
      for (int t0 = 0; t0 < cnt; t0 += 8) {
        float a[8]; int bb[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {             // t >= cnt within the last chunk: multiplier 0, address = start of LD
          int t = t0 + u < 64 ? t0 + u : 63;    // lanes t >= cnt hold multiplier 0 and a valid address: no branch needed
          a[u] = tmw_readlane(ar, t);
          bb[u] = tm_f2i(tmw_readlane(br, t));
        }
        TMW_FOR {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = L[bb[u] + lane];
          float s = acc[TMW_LI];
#pragma unroll
          for (int u = 0; u < 8; u++) s -= a[u] * v[u];
          acc[TMW_LI] = s;
        }
      }
    }
    TMW_TICK2(14);
    float inv = 1.f / tmw_readlane(acc, 0);
    TMW_FOR {
      if (lane <= d) L[K.l_LD + adr + lane] = acc[TMW_LI];
      if (lane == 0) L[K.l_Dinv + i] = inv;
    }
    TMW_SYNC();
    TMW_TICK2(15);
  }
  // rows still hold M' = D L: scale the strict part to the unit-lower L
  TMW_FOR {
    for (int i = lane; i < K.nv; i += 64) {
      int w0 = TMW_W0(i), adr = TMW_ADR(w0), d = TMW_DEPTH(w0);
      float inv = L[K.l_Dinv + i];
      for (int q = 1; q <= d; q++) L[K.l_LD + adr + q] *= inv;
    }
  }
  TMW_SYNC();
}
// root -> leaf pass after tmw_factor(.., rhs): x_i <- x_i / D_i - sum_q L(i,q) x_anc_q(i)   (ancestors are final first)
TM_DEV void tmw_subst_down(WCtx &c, const WLayout &K, int x) {
  float *L = c.L; TMW_LANE_DECL
  TMW_REG(float, t);
  for (int i = 0; i < K.nv; i++) {
    int w0 = TMW_W0(i), w1 = TMW_W1(i), adr = TMW_ADR(w0), d = TMW_DEPTH(w0);
    TMW_FOR { t[TMW_LI] = (lane >= 1 && lane <= d) ? L[K.l_LD + adr + lane] * L[x + tmw_anc(i, lane, w1)] : 0.f; }
    float s = d > 0 ? tmw_sum(t) : 0.f;
    TMW_FOR { if (lane == 0) L[x + i] = L[x + i] * L[K.l_Dinv + i] - s; }
    TMW_SYNC();
  }
}
// in place: strict part of every row of LD becomes the corresponding row of N = L^-1 (same ancestor sparsity), rows
// root -> leaf:  N(k,q) = -( L(k,q) + sum_{m<q} L(k,m) N(anc_m, q-m) ).  Lane q owns column q; lane m pre-computes the
// row address of ancestor m; the loop over m is scalar-broadcast (readlane), one LDS read + one FMA per lane per m.
TM_DEV void tmw_invert_l(WCtx &c, const WLayout &K) {
  float *L = c.L; TMW_LANE_DECL
  TMW_REG(float, x); TMW_REG(float, arow); TMW_REG(float, acc);
  for (int k = 1; k < K.nv; k++) {
    int w0k = TMW_W0(k), w1k = TMW_W1(k), ak = TMW_ADR(w0k), d = TMW_DEPTH(w0k);
    if (d == 0) continue;
    TMW_FOR {
      float xv = (lane >= 1 && lane <= d) ? L[K.l_LD + ak + lane] : 0.f;
      x[TMW_LI] = (lane < d) ? xv : 0.f;   // multipliers L(k,m), m < d only (lane d contributes nothing: zero, no branch)
      acc[TMW_LI] = xv;
      // address such that  L[arow_m + lane]  is entry (lane - m) of ancestor m's row
      int am = tmw_anc(k, lane <= d ? lane : 0, w1k);
      arow[TMW_LI] = tm_i2f(K.l_LD + TMW_ADR(TMW_W0(am)) - lane);
    }
    for (int m0 = 1; m0 < d; m0 += 8) {
      float lm[8]; int ab[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int mc = m0 + u < 64 ? m0 + u : 63;
        lm[u] = tmw_readlane(x, mc);
        ab[u] = tm_f2i(tmw_readlane(arow, mc));
      }
      TMW_FOR {
        float v[8]; int ad[8];
#pragma unroll
        for (int u = 0; u < 8; u++) ad[u] = lane > m0 + u ? ab[u] + lane : K.l_dummy + lane;   // q <= m: not a column of that row
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = L[ad[u]];
        float s = acc[TMW_LI];
#pragma unroll
        for (int u = 0; u < 8; u++) s += lane > m0 + u ? lm[u] * v[u] : 0.f;
        acc[TMW_LI] = s;
      }
    }
    TMW_FOR { if (lane >= 1 && lane <= d) L[K.l_LD + ak + lane] = -acc[TMW_LI]; }
    TMW_SYNC();
  }
}
// ---- register-resident variants for a tree made of CHAINS (WLayout::chains: the rodent, wave_layout.h).
// A chain = N consecutive dofs FIRST.., dof FIRST+k at depth D0+k, whose ancestors above the chain are the trunk dofs
// 0..D0-1 (depth j <-> dof j).  Lane q owns the column of the ancestor at DEPTH q, so that the rows of one root->leaf path
// line up lane by lane: row k of the chain is ONE register (r[k], lane q = M'(k, anc_q)), the trunk rows are 12 more, and a
// rank-1 update / a row of L^-1 is  `s_a = v_readlane(row, j);  target = fma(-s_a, row', target)`  — two instructions per
// (dof, ancestor) pair, no LDS traffic, no address arithmetic, no dependent-load chain.  Everything is unrolled at compile
// time (register arrays cannot be indexed dynamically).  Lane TMW_RL carries the right-hand side of the fused solve.
#define TMW_RL 63
// 1/x: hardware reciprocal (1 ulp) + one Newton step — the pivots are positive, normal numbers (D of an SPD matrix); the IEEE
// division sequence costs ~12 dependent instructions on the critical path of every pivot
TM_DEV float tmw_rcp(float x) {
#ifdef TM_HOST_EMU
  return 1.f / x;
#else
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.f), r, r);
#endif
}
// Lane predicates of the unrolled chain code are COMPILE-TIME lane masks (lane < depth, lane == depth, ..): on the GPU they
// become 64-bit literals moved into an SGPR pair (llvm.amdgcn.inverse.ballot) — no v_cmp, nothing for LICM to hoist and
// keep alive across the substep loop (an earlier version compared `lane` with the ~36 depths and drowned in SGPR spills).
#ifdef TM_HOST_EMU
#define TMW_MASK(m64) (((((unsigned long long)(m64)) >> lane) & 1ull) != 0ull)
#define TMW_SCHED_FENCE() do { } while (0)
#define TMW_PIN(x) do { } while (0)
#else
// A 64-bit literal whose upper 33 bits are all ones (lanes 31..63 set, e.g. "every lane from 18 up") is materialised by hipcc (LLVM 22 /
// ROCm 7.2) as ONE `s_mov_b64 s[..], <32-bit literal>` that it expects the hardware to sign-extend; gfx950 ZERO-extends it and lanes
// 32-63 silently drop out of the mask (tools/micro/literal64.hip shows both).  Such masks get their upper half through an opaque SGPR,
// which makes the compiler emit two s_mov_b32.
TM_DEV unsigned long long tmw_lit64(unsigned long long m64) {
  unsigned hi = (unsigned)(m64 >> 32);
  asm("" : "+s"(hi));
  return ((unsigned long long)hi << 32) | (unsigned)m64;
}
#define TMW_MASK(m64) __builtin_amdgcn_inverse_ballot_w64((((unsigned long long)(m64)) >> 31) == 0x1ffffffffull ? tmw_lit64((unsigned long long)(m64)) : (unsigned long long)(m64))
// keep LLVM from sinking every rank-1 update down to the pivot step of its target row (which turns the elimination into a
// left-looking one whose ~600 multipliers all stay live in SGPRs): an empty volatile asm that "modifies" the target pins
// the FMA to its place in program order
#define TMW_PIN(x) asm volatile("" : "+v"(x))
#define TMW_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define TMW_M_LT(n) ((n) >= 64 ? ~0ull : ((1ull << (n)) - 1ull))
#define TMW_M_EQ(n) (1ull << (n))
#define TMW_M_RANGE(a, b) (TMW_M_LT(b) & ~TMW_M_LT(a))
// Two rows per 64-bit register pair: v_pk_fma_f32 updates both with ONE instruction — the two multipliers come as an SGPR
// pair (two v_readlane), the pivot row is broadcast by op_sel.  1.5 instead of 2 instructions per (dof, ancestor) pair.
#ifdef TM_HOST_EMU
struct tmw_f2 { float x, y; };
TM_DEV tmw_f2 tmw_fnma2(float ax, float ay, float bx, float by, tmw_f2 c) { tmw_f2 r; r.x = c.x - ax * bx; r.y = c.y - ay * by; return r; }
#define TMW_PIN2(v) do { } while (0)
#else
typedef float tmw_f2 __attribute__((ext_vector_type(2)));
TM_DEV tmw_f2 tmw_fnma2(float ax, float ay, float bx, float by, tmw_f2 c) { tmw_f2 a = {ax, ay}, b = {bx, by}; return __builtin_elementwise_fma(-a, b, c); }
#define TMW_PIN2(v) asm volatile("" : "+v"(v))
#endif
#define TMW_ROW(arr, k) (((k) & 1) ? arr[(k) >> 1][TMW_LI].y : arr[(k) >> 1][TMW_LI].x)
#define TMW_SET_ROW(arr, k, v) do { if ((k) & 1) arr[(k) >> 1][TMW_LI].y = (v); else arr[(k) >> 1][TMW_LI].x = (v); } while (0)
// rows [0, CNT) of the paired register array T -= L(k, .) * (pivot row): multipliers are lanes BASE + row of `rs`
// (software-pipelined by hand: the two multipliers of pair p - 1 are read BEFORE the update of pair p is issued.  Read right in front of
// their use, every pair was `v_readlane, v_readlane, s_nop 1, v_pk_fma` through the same SGPR pair — a VALU instruction may read an SGPR
// only two wait states after a VALU instruction wrote it, and the allocator re-used s[0:1] for every pair)
template <typename F>
TM_DEV void tmw_rank1_rows(tmw_f2 (*T)[TMW_NL], const int CNT, const int BASE, const float *rs, const float *rk, F after_top) {
  const int top = (CNT - 1) / 2;
  float alo = tmw_readlane(rs, BASE + 2 * top), ahi = 2 * top + 1 < CNT ? tmw_readlane(rs, BASE + 2 * top + 1) : 0.f;
#pragma unroll
  for (int p = top; p >= 0; p--) {
    float nlo = 0.f, nhi = 0.f;
    if (p > 0) { nlo = tmw_readlane(rs, BASE + 2 * p - 2); nhi = tmw_readlane(rs, BASE + 2 * p - 1); }
    if (2 * p + 1 < CNT) {
      TMW_FOR { T[p][TMW_LI] = tmw_fnma2(alo, ahi, rk[TMW_LI], rk[TMW_LI], T[p][TMW_LI]); TMW_PIN2(T[p][TMW_LI]); }
    } else {
      TMW_FOR { T[p][TMW_LI].x -= alo * rk[TMW_LI]; TMW_PIN2(T[p][TMW_LI]); }
    }
    if (p == top) after_top();
    alo = nlo; ahi = nhi;
  }
}
// EULER = false: plain M (no damping term, no right-hand side)
template <int FIRST, int N, int D0, bool EULER, bool RHS = EULER>
TM_DEV void tmw_rows_load(WCtx &c, const WLayout &K, tmw_f2 (*r)[TMW_NL], float hdamp, int rhs) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  const int adr0 = K.l_M + tmw_chain_madr(FIRST);
  TMW_REG(float, hd);   // hdamp * damping of the chain dof whose DIAGONAL sits in this lane (depth = lane): from LDS (l_hdamp = timestep *
                        // damping; a global load of the model constant here cost one exposed memory latency per chain)
  if (EULER) { TMW_FOR { bool mine = TMW_MASK(TMW_M_RANGE(D0, D0 + N)); hd[TMW_LI] = mine ? L[K.l_hdamp + (mine ? FIRST + lane - D0 : 0)] : 0.f; } }
  TMW_FOR {
#pragma unroll
    for (int k = 0; k < N; k++) {
      const int dk = D0 + k, off = k * D0 + k * (k - 1) / 2 + k;   // Madr(FIRST + k) - Madr(FIRST): rows are stored back to back
      float v = L[adr0 - lane + (off + dk)];   // lanes beyond the row read (and discard) whatever precedes it in LDS
      if (EULER) v += TMW_MASK(TMW_M_EQ(dk)) ? hd[TMW_LI] : 0.f;
      v = TMW_MASK(TMW_M_LT(dk + 1)) ? v : 0.f;
      TMW_SET_ROW(r, k, v);
    }
    if (N & 1) r[N >> 1][TMW_LI].y = 0.f;
  }
  if (RHS) {
    // the right-hand side rides in lane TMW_RL of every row: one vector load of the chain's N values, then per row a v_readlane +
    // a masked move — no per-row LDS access (a uniform-address ds_read per row and, in the factorisation, an exec-masked ds_write
    // per row cost 9 k cycles per substep)
    TMW_REG(float, rhsv);
    TMW_FOR { rhsv[TMW_LI] = L[rhs + FIRST + (lane < N ? lane : N - 1)]; }
#pragma unroll
    for (int k = 0; k < N; k++) {
      float sk = tmw_readlane(rhsv, k);
      TMW_FOR { if (TMW_MASK(TMW_M_EQ(TMW_RL))) TMW_SET_ROW(r, k, sk); }
    }
  }
}
// eliminate the chain leaf -> root: finished rows go to LD (strict part = L; the diagonal word is not written, D^-1 goes to
// Dinv) and the eliminated rhs; the Schur complement lands in the remaining chain rows and in the trunk rows `tr`
// (registers, shared by all chains).  Software-pipelined by hand: as soon as the first update of step k has finished row
// k-1, the pivot chain of step k-1 (readlane -> rcp -> scale) is issued and the remaining updates of step k cover it.
// ---- Schur complement of the leaf chains on the trunk rows as MFMA.  Eliminating chain row k subtracts L(k, i) * M'(k, j) from
// trunk entry (i, j) for all 12 x 12 trunk pairs: over the 61 chain rows that is S -= A^T B with A[k][i] = L(k, trunk_i) and
// B[k][j] = M'(k, trunk_j) — 55 % of all rank-1 updates of the factorisation and a genuine [12 x 61] . [61 x 12] product.
// Four finished rows are packed into one operand register (lanes 16 kk + i = row kk, trunk column i; two v_permlane16_swap + one
// v_permlane32_swap) and go through v_mfma_f32_16x16x4_f32: 7 instructions per four rows instead of 4 x 18 (readlane + pk_fma).
// The accumulator holds C[i][j] at (register i % 4, lane 16 (i / 4) + j) and is subtracted from the trunk rows once, at the end.
// ---- The trunk block in FLOAT64 (round 4).  The trunk diagonal is what is left of the whole body's composite inertia once every limb has been
// eliminated: a small difference of large terms, and in float32 the one place where the leaf -> root L^T D L loses against the dense Cholesky
// that MJX's dense path runs (tests/diagnostics/ldl_vs_cholesky.py: accumulating the 12 trunk rows in float64 takes M^-1 b from 2.9e-6 to
// 1.1e-6 median, 1.7e-5 to 2.9e-6 worst, on the rodent's own matrices; the float32 dense Cholesky: 6.8e-7).  The Schur accumulation is
// v_mfma_f64_16x16x4_f64 on the float32 operands widened at the flush (their products are exact in float64), the twelve trunk rows are then
// eliminated in float64 registers (v_fma_f64, pivots by v_rcp_f64 + two Newton steps) and leave as float32: L, D^-1 and the eliminated rhs are
// stored exactly where the float32 path stores them.  -DTMW_TRUNK_F32 restores the float32 block (the diagnostics' "before" arm).
#ifdef TMW_TRUNK_F32
typedef float tmw_acc_t;
#else
typedef double tmw_acc_t;
#endif
struct TmwSchur {
#ifdef TM_HOST_EMU
  tmw_acc_t C[16][16];
#elif defined(TMW_TRUNK_F32)
  float __attribute__((ext_vector_type(4))) c;       // C[i][j] at (component i % 4, lane 16 (i / 4) + j)
#else
  double __attribute__((ext_vector_type(4))) c;      // C[i][j] at (component i / 4, lane 16 (i % 4) + j): the f64 instruction's own layout
#endif
  float qa[4][TMW_NL], qb[4][TMW_NL];      // operand queue: scaled rows (L) and unscaled rows (M') of the last <= 4 pivots
  float yt[TMW_NL];                        // Euler: lane i = sum_k L(k, trunk_i) y_k - rhs_i  (forward elimination of the rhs on the trunk, running)
};
#ifndef TM_HOST_EMU
TM_DEV float tmw_pack4(float x0, float x1, float x2, float x3) {     // [x0.row0 | x1.row0 | x2.row0 | x3.row0], row = 16 lanes
  auto s01 = __builtin_amdgcn_permlane16_swap(tm_f2i(x0), tm_f2i(x1), false, false);
  auto s23 = __builtin_amdgcn_permlane16_swap(tm_f2i(x2), tm_f2i(x3), false, false);
  auto p = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
  return tm_i2f(p[0]);
}
// lanes of 16-lane row `blk` of x -> row 0 (the other rows of the result are unspecified)
TM_DEV int tmw_row_to0(int x, const int blk) {
  if (blk == 1) return __builtin_amdgcn_permlane16_swap(x, 0, false, false)[1];
  if (blk >= 2) {
    x = __builtin_amdgcn_permlane32_swap(x, 0, false, false)[1];
    if (blk == 3) x = __builtin_amdgcn_permlane16_swap(x, 0, false, false)[1];
  }
  return x;
}
TM_DEV double tmw_row_to0_d(double x, const int blk) {
  long long b; __builtin_memcpy(&b, &x, 8);
  int lo = tmw_row_to0((int)(b & 0xffffffffll), blk), hi = tmw_row_to0((int)(b >> 32), blk);
  b = ((long long)hi << 32) | (unsigned)lo;
  __builtin_memcpy(&x, &b, 8);
  return x;
}
TM_DEV double tmw_readlane_d(const double *v, int src) {
  long long b; __builtin_memcpy(&b, v, 8);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  b = ((long long)hi << 32) | (unsigned)lo;
  double r; __builtin_memcpy(&r, &b, 8);
  return r;
}
// 1 / x in float64 to 2^-46: the float32 hardware reciprocal (1 ulp) as the seed + ONE float64 Newton step (v_rcp_f64 is a quarter-rate instruction
// with a long latency, and it sits on the critical path of every trunk pivot; the pivots are positive, normal numbers well inside float32's range)
TM_DEV double tmw_rcp_d(double x) { double r = (double)__builtin_amdgcn_rcpf((float)x); return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r); }
#else
TM_DEV double tmw_readlane_d(const double *v, int src) { return v[src]; }
TM_DEV double tmw_rcp_d(double x) { return 1.0 / x; }
#endif
// arithmetic of the trunk block in the accumulator's precision
#ifdef TMW_TRUNK_F32
TM_DEV float tmw_acc_readlane(const float *v, int src) { return tmw_readlane(v, src); }
TM_DEV float tmw_acc_rcp(float x) { return tmw_rcp(x); }
TM_DEV float tmw_acc_fnma(float a, float b, float c) { return fmaf(-a, b, c); }
#else
TM_DEV double tmw_acc_readlane(const double *v, int src) { return tmw_readlane_d(v, src); }
TM_DEV double tmw_acc_rcp(double x) { return tmw_rcp_d(x); }
TM_DEV double tmw_acc_fnma(double a, double b, double c) { return __builtin_fma(-a, b, c); }
TM_DEV float tmw_acc_readlane(const float *v, int src) { return tmw_readlane(v, src); }
TM_DEV float tmw_acc_rcp(float x) { return tmw_rcp(x); }
TM_DEV float tmw_acc_fnma(float a, float b, float c) { return fmaf(-a, b, c); }
#endif
// The accumulator STARTS from minus the trunk rows, so that the contributions are taken off the trunk block one group at a
// time, in elimination order (summing all 61 contributions first and subtracting them from the large trunk entries at the end costs a
// factor ~1.4 in accuracy in float32: the trunk diagonal is a small difference of large terms); Euler's right-hand side likewise (S.yt)
template <bool EULER>
TM_DEV void tmw_schur_init(WCtx &c, const WLayout &K, TmwSchur &S, tmw_f2 (*tr)[TMW_NL], int rhs) {
  float *L = c.L; TMW_LANE_DECL
#ifdef TM_HOST_EMU
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) S.C[i][j] = (i < TMW_RODENT_TRUNK) ? -(tmw_acc_t)((i & 1) ? tr[i >> 1][j].y : tr[i >> 1][j].x) : (tmw_acc_t)0;
  for (int q = 0; q < 4; q++) for (int l = 0; l < 64; l++) S.qa[q][l] = S.qb[q][l] = 0.f;
#else
#define TMW_TRROW(i) ((i) < TMW_RODENT_TRUNK ? -(((i) & 1) ? tr[(i) >> 1][0].y : tr[(i) >> 1][0].x) : 0.f)
#ifdef TMW_TRUNK_F32
  S.c[0] = tmw_pack4(TMW_TRROW(0), TMW_TRROW(4), TMW_TRROW(8), TMW_TRROW(12));
  S.c[1] = tmw_pack4(TMW_TRROW(1), TMW_TRROW(5), TMW_TRROW(9), TMW_TRROW(13));
  S.c[2] = tmw_pack4(TMW_TRROW(2), TMW_TRROW(6), TMW_TRROW(10), TMW_TRROW(14));
  S.c[3] = tmw_pack4(TMW_TRROW(3), TMW_TRROW(7), TMW_TRROW(11), TMW_TRROW(15));
#else
  S.c[0] = (double)tmw_pack4(TMW_TRROW(0), TMW_TRROW(1), TMW_TRROW(2), TMW_TRROW(3));
  S.c[1] = (double)tmw_pack4(TMW_TRROW(4), TMW_TRROW(5), TMW_TRROW(6), TMW_TRROW(7));
  S.c[2] = (double)tmw_pack4(TMW_TRROW(8), TMW_TRROW(9), TMW_TRROW(10), TMW_TRROW(11));
  S.c[3] = 0.0;
#endif
#undef TMW_TRROW
  for (int q = 0; q < 4; q++) S.qa[q][0] = S.qb[q][0] = 0.f;
#endif
  TMW_FOR { S.yt[TMW_LI] = EULER ? -L[rhs + (lane < TMW_RODENT_TRUNK ? lane : 0)] : 0.f; }
}
// C += sum over the queued rows of  qa (x) qb  (the queue slots not filled since the last flush must hold zeros)
TM_DEV void tmw_schur_flush(TmwSchur &S) {
#ifdef TM_HOST_EMU
  for (int q = 0; q < 4; q++) for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) S.C[i][j] += (tmw_acc_t)S.qa[q][i] * (tmw_acc_t)S.qb[q][j];
  for (int q = 0; q < 4; q++) for (int l = 0; l < 64; l++) S.qa[q][l] = S.qb[q][l] = 0.f;
#else
  float A = tmw_pack4(S.qa[0][0], S.qa[1][0], S.qa[2][0], S.qa[3][0]), B = tmw_pack4(S.qb[0][0], S.qb[1][0], S.qb[2][0], S.qb[3][0]);
#ifdef TMW_TRUNK_F32
  S.c = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, S.c, 0, 0, 0);
#else
  S.c = __builtin_amdgcn_mfma_f64_16x16x4f64((double)A, (double)B, S.c, 0, 0, 0);
#endif
  for (int q = 0; q < 4; q++) S.qa[q][0] = S.qb[q][0] = 0.f;
#endif
}
// trunk rows t[i] <- -C[i][.] in lanes 0..15 (and, Euler, the eliminated rhs -yt_i in lane TMW_RL; every other lane: zero)
template <bool EULER>
TM_DEV void tmw_schur_apply(WCtx &c, TmwSchur &S, tmw_acc_t (*t)[TMW_NL]) {
  TMW_LANE_DECL
#pragma unroll
  for (int i = 0; i < TMW_RODENT_TRUNK; i++) {
#ifdef TM_HOST_EMU
    TMW_FOR { tmw_acc_t v = 0; if (lane < 16) v = -S.C[i][lane]; if (EULER && lane == TMW_RL) v = -(tmw_acc_t)S.yt[i]; t[i][TMW_LI] = v; }
#else
#ifdef TMW_TRUNK_F32
    float x = tm_i2f(tmw_row_to0(tm_f2i(S.c[i & 3]), i / 4));
#else
    double x = tmw_row_to0_d(S.c[i / 4], i & 3);
#endif
    tmw_acc_t v = TMW_MASK(TMW_M_LT(16)) ? -x : (tmw_acc_t)0;
    if (EULER) { float y = tmw_readlane(S.yt, i); v = TMW_MASK(TMW_M_EQ(TMW_RL)) ? (tmw_acc_t)(-y) : v; }
    t[i][0] = v;
#endif
  }
}
// the twelve trunk rows, leaf -> root, in the accumulator's precision: same outputs as tmw_rows_factor<0, TRUNK, 0, EULER> (strict part of L,
// D^-1, the eliminated rhs — all stored as float32)
template <bool EULER, typename AT>
TM_DEV void tmw_trunk_factor(WCtx &c, const WLayout &K, AT (*t)[TMW_NL], int rhs) {
  float *L = c.L; TMW_LANE_DECL
  constexpr int N = TMW_RODENT_TRUNK;
  TMW_REG(float, dv); TMW_REG(float, yv);
  AT rs[2][TMW_NL], inv[2];
  TMW_FOR { dv[TMW_LI] = 0.f; yv[TMW_LI] = 0.f; }
  // software-pipelined like tmw_rows_factor: as soon as the first update of step k has finished row k - 1, that row's pivot chain (readlane ->
  // reciprocal -> scale) is issued and the remaining updates of step k cover its latency
  inv[(N - 1) & 1] = tmw_acc_rcp(tmw_acc_readlane(t[N - 1], N - 1));
  TMW_FOR { rs[(N - 1) & 1][TMW_LI] = t[N - 1][TMW_LI] * inv[(N - 1) & 1]; }
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    const int off = k * (k - 1) / 2 + k, b = k & 1;          // Madr(k) of the trunk chain (depth = dof)
    if (k > 0) {
      { const AT a = tmw_acc_readlane(rs[b], k - 1); TMW_FOR { t[k - 1][TMW_LI] = tmw_acc_fnma(a, t[k][TMW_LI], t[k - 1][TMW_LI]); } }
      inv[b ^ 1] = tmw_acc_rcp(tmw_acc_readlane(t[k - 1], k - 1));
      TMW_FOR { rs[b ^ 1][TMW_LI] = t[k - 1][TMW_LI] * inv[b ^ 1]; }
    }
#pragma unroll
    for (int i = k - 2; i >= 0; i--) { const AT a = tmw_acc_readlane(rs[b], i); TMW_FOR { t[i][TMW_LI] = tmw_acc_fnma(a, t[k][TMW_LI], t[i][TMW_LI]); } }
    float yk = 0.f;
    if (EULER) yk = (float)tmw_acc_readlane(t[k], TMW_RL);
    TMW_FOR {
      if (EULER) yv[TMW_LI] = TMW_MASK(TMW_M_EQ(k)) ? yk : yv[TMW_LI];
      if (k > 0 && TMW_MASK(TMW_M_LT(k))) L[K.l_LD - lane + (off + k)] = (float)rs[b][TMW_LI];
      dv[TMW_LI] = TMW_MASK(TMW_M_EQ(k)) ? (float)inv[b] : dv[TMW_LI];
    }
  }
  TMW_FOR {
    if (TMW_MASK(TMW_M_LT(N))) { L[K.l_Dinv + lane] = dv[TMW_LI]; if (EULER) L[rhs + lane] = yv[TMW_LI]; }
  }
}

template <int FIRST, int N, int D0, bool EULER>
TM_DEV void tmw_rows_factor(WCtx &c, const WLayout &K, tmw_f2 (*r)[TMW_NL], TmwSchur *S, int rhs) {
  float *L = c.L; TMW_LANE_DECL
  const int adr0 = K.l_LD + tmw_chain_madr(FIRST);
  constexpr int Q0 = D0 > 0 ? tmw_chain_rows_before(FIRST) : 0;      // rows queued before this chain (mod 4 = first queue slot)
  float rs[2][TMW_NL], rk[2][TMW_NL], inv[2];
  TMW_REG(float, dv);
  TMW_REG(float, yv);       // Euler: lane k = eliminated right-hand side of chain row k, stored once behind the loop
  TMW_FOR { yv[TMW_LI] = 0.f; rk[(N - 1) & 1][TMW_LI] = TMW_ROW(r, N - 1); }
  inv[(N - 1) & 1] = tmw_rcp(tmw_readlane(rk[(N - 1) & 1], D0 + N - 1));
  TMW_FOR { rs[(N - 1) & 1][TMW_LI] = rk[(N - 1) & 1][TMW_LI] * inv[(N - 1) & 1]; }
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    const int dk = D0 + k, off = k * D0 + k * (k - 1) / 2 + k, b = k & 1;
    if (k > 0) {
      tmw_rank1_rows(r, k, D0, rs[b], rk[b], [&]() {
        TMW_FOR { rk[b ^ 1][TMW_LI] = TMW_ROW(r, k - 1); }
        inv[b ^ 1] = tmw_rcp(tmw_readlane(rk[b ^ 1], dk - 1));
        TMW_FOR { rs[b ^ 1][TMW_LI] = rk[b ^ 1][TMW_LI] * inv[b ^ 1]; TMW_PIN(rs[b ^ 1][TMW_LI]); }
      });
    }
    float yk = 0.f;
    if (EULER) { yk = tmw_readlane(rk[b], TMW_RL); TMW_FOR { yv[TMW_LI] = TMW_MASK(TMW_M_EQ(k)) ? yk : yv[TMW_LI]; } }
    if (D0 > 0) {     // trunk part of the finished row -> operand queue of the Schur MFMA (lanes >= D0 are not trunk columns: zero them)
      const int slot = (Q0 + (N - 1 - k)) & 3;
      TMW_FOR {
        float a = D0 < 16 ? (TMW_MASK(TMW_M_LT(D0)) ? rs[b][TMW_LI] : 0.f) : rs[b][TMW_LI];
        S->qa[slot][TMW_LI] = a;
        S->qb[slot][TMW_LI] = D0 < 16 ? (TMW_MASK(TMW_M_LT(D0)) ? rk[b][TMW_LI] : 0.f) : rk[b][TMW_LI];
      }
      if (EULER) { TMW_FOR { S->yt[TMW_LI] += S->qa[slot][TMW_LI] * yk; } }
      if (slot == 3) tmw_schur_flush(*S);
    }
    TMW_FOR {
      if (dk > 0 && TMW_MASK(TMW_M_LT(dk))) L[adr0 - lane + (off + dk)] = rs[b][TMW_LI];
      dv[TMW_LI] = TMW_MASK(TMW_M_EQ(dk)) ? inv[b] : dv[TMW_LI];
    }
  }
  TMW_FOR {
    if (TMW_MASK(TMW_M_RANGE(D0, D0 + N))) L[K.l_Dinv + FIRST - D0 + lane] = dv[TMW_LI];
    if (EULER && TMW_MASK(TMW_M_LT(N))) L[rhs + FIRST + lane] = yv[TMW_LI];
  }
}
template <int FIRST, int N, int D0, bool EULER>
TM_DEV void tmw_chain_factor(WCtx &c, const WLayout &K, TmwSchur *S, float hdamp, int rhs) {
  tmw_f2 r[(N + 1) / 2][TMW_NL];
  tmw_rows_load<FIRST, N, D0, EULER>(c, K, r, hdamp, rhs);
  tmw_rows_factor<FIRST, N, D0, EULER>(c, K, r, S, rhs);
}
// LD <- L^T D L of M (EULER = false) or of M + hdamp diag(damping) with the rhs eliminated alongside (EULER = true),
// read straight from l_M; same outputs as tmw_factor except that the diagonal words of LD are left alone
template <bool EULER>
TM_DEV void tmw_factor_chains(WCtx &c, const WLayout &K, float hdamp, int rhs) {
  TmwSchur S;
  {
    tmw_f2 tr[(TMW_RODENT_TRUNK + 1) / 2][TMW_NL];
    tmw_rows_load<0, TMW_RODENT_TRUNK, 0, EULER, false>(c, K, tr, hdamp, rhs);
    tmw_schur_init<EULER>(c, K, S, tr, rhs);
  }
#define TMW_X(first, n, d0) tmw_chain_factor<first, n, d0, EULER>(c, K, &S, hdamp, rhs);
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  tmw_schur_flush(S);                       // the last, partly filled group (unused slots hold zeros)
  tmw_acc_t t[TMW_RODENT_TRUNK][TMW_NL];
  tmw_schur_apply<EULER>(c, S, t);
#if defined(TMW_TRUNK_ELIM_F32) && !defined(TMW_TRUNK_F32)      // diagnostic arm (tests/diagnostics/trunk_f64.py): float64 Schur accumulation only
  { float tf[TMW_RODENT_TRUNK][TMW_NL];
    TMW_FOR { for (int i = 0; i < TMW_RODENT_TRUNK; i++) tf[i][TMW_LI] = (float)t[i][TMW_LI]; }
    tmw_trunk_factor<EULER, float>(c, K, tf, rhs); }
#else
  tmw_trunk_factor<EULER, tmw_acc_t>(c, K, t, rhs);
#endif
  TMW_SYNC();
}
// rows of N = L^-1, root -> leaf:  N(k,:) = e_k - sum_{j < depth_k} L(k, anc_j) N(anc_j, :).  `tn`: the finished trunk rows
// (with their unit diagonal); chain rows are kept in `n` the same way.  Lanes beyond a row's depth hold exact zeros.
// acc -= L(k, rows [0, CNT) of S): multipliers are lanes BASE + row of `l`; two rows per v_pk_fma_f32
TM_DEV void tmw_rows_accum(tmw_f2 &acc, const tmw_f2 (*S)[TMW_NL], const int CNT, const int BASE, const float *l, int li) {
#pragma unroll
  for (int p = 0; 2 * p < CNT; p++) {
    float alo = tmw_readlane(l, BASE + 2 * p), ahi = 2 * p + 1 < CNT ? tmw_readlane(l, BASE + 2 * p + 1) : 0.f;
    acc = tmw_fnma2(alo, ahi, S[p][li].x, S[p][li].y, acc);
  }
}
// Trunk-ancestor part of the rows of L^-1 of a whole leaf chain as MFMA:  P[k][q] = sum_{j < D0} L(k, trunk_j) N(trunk_j, q).
// A[i][kk] = L(k0 + i, trunk 4 m + kk) is read from LDS directly in operand layout (lane 16 kk + i), B = four trunk rows of
// L^-1 packed into one register (tmw_pack4), 16 chain rows per accumulator; the rows of P are then pulled out of the accumulator
// (register i % 4, lanes 16 (i / 4) ..) with at most two v_permlane swaps each — instead of 12 readlane + 6 pk_fma per row.
template <int FIRST, int N, int D0>
TM_DEV void tmw_chain_trunk_products(WCtx &c, const WLayout &K, const float (*Bp)[TMW_NL], const tmw_f2 (*tn)[TMW_NL], float (*P)[TMW_NL]) {
  float *L = c.L; TMW_LANE_DECL
  const int adr0 = K.l_LD + tmw_chain_madr(FIRST);
#ifdef TM_HOST_EMU
  (void)Bp;
  for (int k = 0; k < N; k++)
    TMW_FOR {
      float s = 0.f;
      if (lane < D0) for (int j = 0; j < D0; j++) s += L[adr0 + k * D0 + k * (k - 1) / 2 + k + (D0 + k) - j] * ((j & 1) ? tn[j >> 1][lane].y : tn[j >> 1][lane].x);
      P[k][TMW_LI] = s;
    }
#else
  (void)tn;
  typedef float __attribute__((ext_vector_type(4))) f4;
#pragma unroll
  for (int k0 = 0; k0 < N; k0 += 16) {
    const int kq = k0 + (lane & 15), kk = lane >> 4;
    const bool rowok = kq < N;
    const int kc = rowok ? kq : N - 1;
    const int base = adr0 + kc * D0 + kc * (kc - 1) / 2 + kc + (D0 + kc) - kk;      // entry of trunk column kk; column 4 m + kk is 4 m words below
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; 4 * m < D0; m++) {
      float a = L[base - 4 * m];
      a = (rowok && 4 * m + kk < D0) ? a : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bp[m][0], acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (k0 + i >= N) break;
      float x = acc[i & 3];
      if ((i >> 2) == 1) x = tm_i2f(__builtin_amdgcn_permlane16_swap(tm_f2i(x), 0, false, false)[1]);
      else if ((i >> 2) >= 2) {
        x = tm_i2f(__builtin_amdgcn_permlane32_swap(tm_f2i(x), 0, false, false)[1]);
        if ((i >> 2) == 3) x = tm_i2f(__builtin_amdgcn_permlane16_swap(tm_f2i(x), 0, false, false)[1]);
      }
      P[k0 + i][0] = TMW_MASK(TMW_M_LT(D0)) ? x : 0.f;
    }
  }
#endif
}
template <int FIRST, int N, int D0>
TM_DEV void tmw_rows_invert(WCtx &c, const WLayout &K, tmw_f2 (*n)[TMW_NL], tmw_f2 (*tn)[TMW_NL], const float (*Bp)[TMW_NL]) {
  float *L = c.L; TMW_LANE_DECL
  const int adr0 = K.l_LD + tmw_chain_madr(FIRST);
  float l[N][TMW_NL], P[D0 > 0 ? N : 1][TMW_NL];
  tmw_f2 acc[TMW_NL];
  // all rows of L first: the loads of later rows must not queue behind the stores of the finished rows of N
  TMW_FOR {
#pragma unroll
    for (int k = 0; k < N; k++) {
      const int dk = D0 + k, off = k * D0 + k * (k - 1) / 2 + k;
      float v = L[adr0 - lane + (off + dk)];
      l[k][TMW_LI] = TMW_MASK(TMW_M_LT(dk)) ? v : 0.f;
    }
    // (rows not yet computed enter the odd tail of a pair with a zero multiplier: they must hold numbers)
#pragma unroll
    for (int p = 0; p < (N + 1) / 2; p++) { n[p][TMW_LI].x = 0.f; n[p][TMW_LI].y = 0.f; }
  }
  if (D0 > 0) tmw_chain_trunk_products<FIRST, N, D0>(c, K, Bp, tn, P);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int dk = D0 + k, off = k * D0 + k * (k - 1) / 2 + k;
    TMW_FOR { acc[TMW_LI].x = (TMW_MASK(TMW_M_EQ(dk)) ? 1.f : 0.f) - (D0 > 0 ? P[D0 > 0 ? k : 0][TMW_LI] : 0.f); acc[TMW_LI].y = 0.f; }
    if (k > 0) { TMW_FOR { tmw_rows_accum(acc[TMW_LI], n, k, D0, l[k], TMW_LI); } }
    TMW_FOR {
      float v = acc[TMW_LI].x + acc[TMW_LI].y;
      TMW_SET_ROW(n, k, v);
      // (lane dk = the row's unit diagonal: stored too, kept: N = L^-1 with its diagonal)
      if (TMW_MASK(TMW_M_LT(dk + 1))) L[adr0 - lane + (off + dk)] = v;
    }
  }
}
template <int FIRST, int N, int D0>
TM_DEV void tmw_chain_invert(WCtx &c, const WLayout &K, tmw_f2 (*tn)[TMW_NL], const float (*Bp)[TMW_NL]) {
  tmw_f2 n[(N + 1) / 2][TMW_NL];
  tmw_rows_invert<FIRST, N, D0>(c, K, n, tn, Bp);
}
TM_DEV void tmw_invert_chains(WCtx &c, const WLayout &K) {
  tmw_f2 tn[(TMW_RODENT_TRUNK + 1) / 2][TMW_NL];
  float Bp[(TMW_RODENT_TRUNK + 3) / 4][TMW_NL];
  tmw_rows_invert<0, TMW_RODENT_TRUNK, 0>(c, K, tn, tn, Bp);
#ifndef TM_HOST_EMU
  // B operands of the trunk products: four finished trunk rows of L^-1 per register (exact zeros beyond each row's depth)
#define TMW_TN(i) ((i) < TMW_RODENT_TRUNK ? (((i) & 1) ? tn[(i) >> 1][0].y : tn[(i) >> 1][0].x) : 0.f)
#pragma unroll
  for (int m = 0; m < (TMW_RODENT_TRUNK + 3) / 4; m++) Bp[m][0] = tmw_pack4(TMW_TN(4 * m), TMW_TN(4 * m + 1), TMW_TN(4 * m + 2), TMW_TN(4 * m + 3));
#undef TMW_TN
#endif
#define TMW_X(first, n, d0) tmw_chain_invert<first, n, d0>(c, K, tn, Bp);
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  TMW_SYNC();
}
// root -> leaf pass after tmw_factor_chains<true>:  x_k = y_k / D_k - sum_j L(k, anc_j) x_anc_j.  TRANSPOSED lane layout:
// lane k = row k of the chain, the loop runs over the ancestor depth q, the already final x_q is a scalar broadcast
// (v_readlane) — one load + one FMA per (chain, depth) for all rows at once and no wave reduction (the lane = depth
// layout needed one 6-step DPP reduction per dof).  `xt`: lane j = solution of trunk dof j.
template <int FIRST, int N, int D0>
TM_DEV void tmw_rows_subst(WCtx &c, const WLayout &K, const float *xt, float *z, int x) {
  float *L = c.L; TMW_LANE_DECL
  constexpr int QMAX = D0 + N - 1;
  TMW_REG(float, bv);
  TMW_FOR {
    int k = lane < N ? lane : N - 1;                                    // lanes beyond the chain shadow its last row
    int base = K.l_LD + tmw_chain_madr(FIRST) + k * D0 + k * (k - 1) / 2 + k + D0 + k - QMAX;   // entry of depth q: base + (QMAX - q)
    bv[TMW_LI] = tm_i2f(base);
    z[TMW_LI] = L[x + FIRST + k] * L[K.l_Dinv + FIRST + k];
  }
#pragma unroll
  for (int q = 0; q < D0; q++) {
    float xq = tmw_readlane(xt, q);
    TMW_FOR { z[TMW_LI] -= L[tm_f2i(bv[TMW_LI]) + (QMAX - q)] * xq; }
  }
#pragma unroll
  for (int q = D0; q < QMAX; q++) {
    float xq = tmw_readlane(z, q - D0);                                 // row q - D0 is final: rows below it take its term
    TMW_FOR { float v = L[tm_f2i(bv[TMW_LI]) + (QMAX - q)]; z[TMW_LI] = TMW_MASK(TMW_M_RANGE(q - D0 + 1, N)) ? z[TMW_LI] - v * xq : z[TMW_LI]; }       // (rows q - D0 + 1 .. N - 1: lanes beyond the chain are never stored, and a mask with its upper half set costs an SGPR pair — tmw_lit64)
  }
  TMW_FOR { if (TMW_MASK(TMW_M_LT(N))) L[x + FIRST + lane] = z[TMW_LI]; }
}
TM_DEV void tmw_subst_chains(WCtx &c, const WLayout &K, int x) {
  TMW_LANE_DECL
  TMW_REG(float, xt); TMW_REG(float, z);
  TMW_FOR { xt[TMW_LI] = 0.f; }
  tmw_rows_subst<0, TMW_RODENT_TRUNK, 0>(c, K, xt, xt, x);
#define TMW_X(first, n, d0) tmw_rows_subst<first, n, d0>(c, K, xt, z, x);
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  TMW_SYNC();
}
// column part of a product with ancestor-sparse storage `A`:  c_i = sum_{k descendant of i} A(k,i) x_k  for every dof, rows
// streamed once through a register in the lane = depth layout (per row: one ds_read, one readlane of x_k, one masked FMA;
// the lane-per-column loop it replaces walked up to 72 descendants with two dependent LDS reads each).
//   SOLVE = false:  out_i = c_i                          (out may not alias x)
//   SOLVE = true :  out_i = (x_i + c_i) * Dinv_i         (first half of M^-1 x with A = L^-1; out == x is fine: every x is
//                                                         read before the first store)
template <int FIRST, int N, int D0>
TM_DEV void tmw_rows_colacc(WCtx &c, const WLayout &K, int A, const float *xv, float *acc) {
  float *L = c.L; TMW_LANE_DECL
  const int adr0 = A + tmw_chain_madr(FIRST);
#pragma unroll
  for (int k = N - 1; k >= 1 - (D0 > 0); k--) {
    const int dk = D0 + k, off = k * D0 + k * (k - 1) / 2 + k;
    float xk = tmw_readlane(xv, dk);
    TMW_FOR { float v = L[adr0 - lane + (off + dk)]; acc[TMW_LI] += (TMW_MASK(TMW_M_LT(dk)) ? v : 0.f) * xk; }
  }
}
// all loads of x first, all stores last: the 73 row loads in between have no LDS store to be ordered against
template <bool SOLVE>
TM_DEV void tmw_colpart_chains(WCtx &c, const WLayout &K, int A, int x, int out) {
  float *L = c.L; TMW_LANE_DECL
  float xv[TMW_RODENT_NCHAIN + 1][TMW_NL], acc[TMW_RODENT_NCHAIN + 1][TMW_NL];
  { int ci = 0;
#define TMW_X(first, n, d0) TMW_FOR { bool mine = TMW_MASK(TMW_M_RANGE(d0, d0 + n)); xv[ci][TMW_LI] = mine ? L[x + (mine ? first - d0 + lane : 0)] : 0.f; acc[ci][TMW_LI] = 0.f; } ci++;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
  TMW_X(0, TMW_RODENT_TRUNK, 0)
#undef TMW_X
  }
  { int ci = 0;
#define TMW_X(first, n, d0) tmw_rows_colacc<first, n, d0>(c, K, A, xv[ci], acc[ci]); TMW_FOR { acc[TMW_RODENT_NCHAIN][TMW_LI] += TMW_MASK(TMW_M_LT(d0)) ? acc[ci][TMW_LI] : 0.f; } ci++;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  tmw_rows_colacc<0, TMW_RODENT_TRUNK, 0>(c, K, A, xv[ci], acc[ci]); }
  { int ci = 0;
#define TMW_STORE(first, n, d0) TMW_FOR { if (TMW_MASK(TMW_M_RANGE(d0, d0 + n))) { float v = acc[ci][TMW_LI]; if (SOLVE) v = (xv[ci][TMW_LI] + v) * L[K.l_Dinv + first - d0 + lane]; L[out + first - d0 + lane] = v; } } ci++;
  TMW_RODENT_LEAF_CHAINS(TMW_STORE)
  TMW_STORE(0, TMW_RODENT_TRUNK, 0)
#undef TMW_STORE
  }
  TMW_SYNC();
}
// Lean sparse row / column products shared by tmw_solve and tmw_mul_m.  Row i of the ancestor-sparse storage `A` is
// [A(i,i), A(i,i-1), .., A(i,chain_start), A(i,jump), .., A(i,0)]; column i is { A(k,i) : k = i+1 .. i+ndesc },
// entry (k,i) sitting at Mend_k - depth_i.  No per-iteration ancestor arithmetic: two pointer runs for the row, one table
// word per descendant for the column.  `diag`: include q = 0.
TM_DEV float tmw_row_dot(const float *L, const WLayout &K, int A, int x, int i, bool diag) {
  int w0 = tm_f2i(L[K.l_tdof + 2 * i]), w1 = tm_f2i(L[K.l_tdof + 2 * i + 1]);
  int d = TMW_DEPTH(w0), adr = TMW_ADR(w0), r = i - (w1 & 0xff), jp1 = (w1 >> 8) & 0xff;
  const float *Ap = L + A + adr, *xp = L + x;
  float acc = diag ? Ap[0] * xp[i] : 0.f;
#pragma unroll 4
  for (int q = 1; q <= r; q++) acc += Ap[q] * xp[i - q];
  const float *Aq = Ap + r + 1;
#pragma unroll 4
  for (int t = 0; t < d - r; t++) acc += Aq[t] * xp[jp1 - 1 - t];
  return acc;
}
TM_DEV float tmw_col_dot(const float *L, const WLayout &K, int A, int x, int i) {
  int w0 = tm_f2i(L[K.l_tdof + 2 * i]), nd = (tm_f2i(L[K.l_tdof + 2 * i + 1]) >> 16) & 0xff, d = TMW_DEPTH(w0);
  const float *Ap = L + A - d, *xp = L + x + i + 1;
  const float *tw = L + K.l_tdof + 2 * (i + 1);
  float acc = 0.f;
#pragma unroll 4
  for (int t = 0; t < nd; t++) acc += Ap[TMW_MEND(tm_f2i(tw[2 * t]))] * xp[t];
  return acc;
}
// row part for the chain layout, lane = dof: the two pointer runs of tmw_row_dot (ancestors inside the dof's own chain, then
// on the trunk) as counted loops with a trip count common to the whole slot and a predicate per entry — no divergent loop
// exits, one wait per 6 entries.  The bounds are made opaque so that the loops stay loops: fully unrolled, every
// `q <= r(lane)` would be a loop-invariant lane mask that LLVM hoists out of the substep loop and spills.
#ifdef TM_HOST_EMU
TM_DEV int tmw_opaque_s(int x) { return x; }
#else
TM_DEV int tmw_opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
#endif
// both dof slots of a lane (i and i + 64) advance in the same loops, so that their loads share the LDS round trips
template <int MAXR0, int MAXR1, int MAXT>
TM_DEV void tmw_row_runs2(const float *L, const WLayout &K, int A, int x, int lane, bool diag, float &z0, float &z1, int tp0, int tp1) {
  const int i0 = lane, i1 = lane + 64 < K.nv ? lane + 64 : 64;
  int d0, r0, j0, d1, r1, j1, adr0, adr1;
  if (K.lean) {       // the packed words come in registers (WCtx::tp0 / tp1)
    d0 = TMW_TP_DEPTH(tp0); r0 = TMW_TP_RUN(tp0); j0 = d0 - r0; adr0 = TMW_TP_MEND(tp0) - d0;
    d1 = TMW_TP_DEPTH(tp1); r1 = TMW_TP_RUN(tp1); j1 = d1 - r1; adr1 = TMW_TP_MEND(tp1) - d1;
  } else {
    int w00 = tm_f2i(L[K.l_tdof + 2 * i0]), w01 = tm_f2i(L[K.l_tdof + 2 * i0 + 1]);
    int w10 = tm_f2i(L[K.l_tdof + 2 * i1]), w11 = tm_f2i(L[K.l_tdof + 2 * i1 + 1]);
    d0 = TMW_DEPTH(w00); r0 = i0 - (w01 & 0xff); j0 = (w01 >> 8) & 0xff; adr0 = TMW_ADR(w00);
    d1 = TMW_DEPTH(w10); r1 = i1 - (w11 & 0xff); j1 = (w11 >> 8) & 0xff; adr1 = TMW_ADR(w10);
  }
  const float *A0 = L + A + adr0, *x0 = L + x + i0, *A1 = L + A + adr1, *x1 = L + x + i1;
  float a0 = diag ? A0[0] * x0[0] : 0.f, a1 = diag ? A1[0] * x1[0] : 0.f;
  const int maxr0 = tmw_opaque_s(MAXR0), maxr1 = tmw_opaque_s(MAXR1), maxt = tmw_opaque_s(MAXT);
#pragma unroll 4
  for (int q = 1; q <= maxr1; q++) {
    float u0 = A0[q], v0 = x0[-q], u1 = A1[q], v1 = x1[-q];
    a0 = fmaf(q <= r0 ? u0 : 0.f, v0, a0); a1 = fmaf(q <= r1 ? u1 : 0.f, v1, a1);      // (select, fma: one instruction less than product, select, add)
  }
#pragma unroll 8
  for (int q = maxr1 + 1; q <= maxr0; q++) { float u0 = A0[q], v0 = x0[-q]; a0 = fmaf(q <= r0 ? u0 : 0.f, v0, a0); }
  const float *B0 = A0 + r0 + 1, *y0 = L + x + j0 - 1, *B1 = A1 + r1 + 1, *y1 = L + x + j1 - 1;
  const int n0 = d0 - r0, n1 = d1 - r1;
#pragma unroll 4
  for (int t = 0; t < maxt; t++) {
    float u0 = B0[t], v0 = y0[-t], u1 = B1[t], v1 = y1[-t];
    a0 = fmaf(t < n0 ? u0 : 0.f, v0, a0); a1 = fmaf(t < n1 ? u1 : 0.f, v1, a1);
  }
  z0 = a0; z1 = a1;
}
// (A x)_i over the ancestors of i (+ diagonal) for every dof, kept in two lane registers
TM_DEV void tmw_rowpart_chains(WCtx &c, const WLayout &K, int A, int x, bool diag, float *z0, float *z1) {
  float *L = c.L; TMW_LANE_DECL
  TMW_FOR { tmw_row_runs2<tmw_chain_maxrun(0, 64), tmw_chain_maxrun(64, 73), TMW_RODENT_TRUNK>(L, K, A, x, lane, diag, z0[TMW_LI], z1[TMW_LI], tmw_opaque_v(c.tp0[TMW_LI]), tmw_opaque_v(c.tp1[TMW_LI])); }
}
// (the same mat-vecs on the matrix cores were measured SLOWER than the vector-ALU versions below in round 4 — DESIGN.md "K2 in detail"; that
// experiment, csrc/wave_matvec.h, left the tree in round 5 with the LDS table it read: git history, commit a70fc19)
// x <- M^-1 x using N = L^-1:  x = N D^-1 N^T x   (two sparse mat-vecs; `x` is an LDS vector offset, `tmp` a free one)
TM_DEV void tmw_solve(WCtx &c, const WLayout &K, int x, int tmp) {
  float *L = c.L; TMW_LANE_DECL
  TMW_REG(float, z0); TMW_REG(float, z1);
  if (K.chains) {
    TMW_TICK2(15);
    tmw_colpart_chains<true>(c, K, K.l_LD, x, x);
    TMW_TICK2(26);
    tmw_rowpart_chains(c, K, K.l_LD, x, false, z0, z1);
    TMW_FOR { z0[TMW_LI] += L[x + lane]; if (lane + 64 < K.nv) z1[TMW_LI] += L[x + lane + 64]; }
    TMW_SYNC();
    TMW_FOR { L[x + lane] = z0[TMW_LI]; if (lane + 64 < K.nv) L[x + lane + 64] = z1[TMW_LI]; }
    TMW_SYNC();
    return;
  }
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      (slot ? z1 : z0)[TMW_LI] = (L[x + i] + tmw_col_dot(L, K, K.l_LD, x, i)) * L[K.l_Dinv + i];
    }
  }
  TMW_SYNC();
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) { int i = lane + 64 * slot; if (i < K.nv) L[x + i] = (slot ? z1 : z0)[TMW_LI]; }
  }
  TMW_SYNC();
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      (slot ? z1 : z0)[TMW_LI] = L[x + i] + tmw_row_dot(L, K, K.l_LD, x, i, false);
    }
  }
  TMW_SYNC();
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) { int i = lane + 64 * slot; if (i < K.nv) L[x + i] = (slot ? z1 : z0)[TMW_LI]; }
  }
  TMW_SYNC();
}
// y = M x
TM_DEV void tmw_mul_m(WCtx &c, const WLayout &K, int x, int y) {
  float *L = c.L; TMW_LANE_DECL
  if (K.chains) {
    TMW_REG(float, z0); TMW_REG(float, z1);
    TMW_TICK2(15);
    tmw_rowpart_chains(c, K, K.l_M, x, true, z0, z1);
    TMW_SCHED_FENCE();
    tmw_colpart_chains<false>(c, K, K.l_M, x, y);
    TMW_FOR { L[y + lane] += z0[TMW_LI]; if (lane + 64 < K.nv) L[y + lane + 64] += z1[TMW_LI]; }
    TMW_SYNC();
    TMW_TICK2(24);
  } else {
    TMW_FOR {
      for (int i = lane; i < K.nv; i += 64) L[y + i] = tmw_row_dot(L, K, K.l_M, x, i, true) + tmw_col_dot(L, K, K.l_M, x, i);
    }
    TMW_SYNC();
  }
}

// ------------------------------------------------------------------------------------------ matrix-free J
// J v in two stages (no barrier inside a stage, so a caller can put independent work next to it):
//   stage 1: spatial velocity of each paw body -> l_sv;   stage 2 (after a barrier): one lane per active row
TM_DEV void tmw_jmul_stage1(WCtx &c, const WLayout &K, int v) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  // no penetrating contact (a wave-uniform fact of the env's substep, ~ 2 of 3 substeps): stage 2 reads l_sv for contact rows only
  if (c.nact == c.nla) return;
  TMW_FOR {
    if (lane < K.ngroup * 6) {  // spatial velocity of each paw body, one lane per (group, component)
      int g = lane / 6, k = lane - g * 6, ld, d, run, jp1;
      if (K.lean) { const int gp = tm_f2i(L[K.l_tgrp + g]); ld = (gp & 0xff) == 0xff ? -1 : (gp & 0xff); d = (gp >> 8) & 0xff; run = (gp >> 16) & 0xff; jp1 = d - run; }
      else { ld = ((const signed char *)(L + K.l_tgrp))[g]; const int w1 = TMW_W1(ld >= 0 ? ld : 0); d = TMW_DEPTH(TMW_W0(ld >= 0 ? ld : 0)); run = ld - (w1 & 0xff); jp1 = (w1 >> 8) & 0xff; }
      float s = 0.f;
      if (ld >= 0) {
#pragma unroll 4
        for (int q = 0; q <= d; q++) { int i = tmw_anc_r(ld, q, run, jp1); s += L[K.l_cdof + i * 6 + k] * L[v + i]; }
      }
      L[K.l_sv + lane] = s;
    }
  }
}
TM_DEV void tmw_jmul_stage2(WCtx &c, const WLayout &K, int v, int out) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  // one lane per ACTIVE row (compact row space): a violated limit is one-hot, a pyramid edge of an active contact is
  // n.vel +- mu t.vel of the paw body's point velocity
  TMW_FOR {
    const unsigned char *rm = TMW_ROWMAP(K);
    for (int kr = lane; kr < c.nact; kr += 64) {
      int r = rm[kr];
      float o;
      if (kr < c.nla) {       // limit rows come first; their map byte is dof | (negative side) << 7 (tmw_make_constraint)
        o = ((r & 0x80) ? -1.f : 1.f) * L[v + (r & 0x7f)];
      } else {
        int cc = (r - K.nlim) >> 2, e = (r - K.nlim) & 3;
        const float *sv = L + K.l_sv + TMW_CONGRP(K)[cc] * 6, *off = L + K.l_con_off + cc * 3;
        float fr[9];
        tmw_get_con_frame(L, K, cc, fr);
        tm_cross(fr + 6, fr, fr + 3);
        float cr[3], vel[3];
        tm_cross(cr, sv, off);
        for (int k = 0; k < 3; k++) vel[k] = sv[3 + k] + cr[k];
        float a0 = tm_dot3(fr, vel), at = tm_dot3(fr + 3 + 3 * (e >> 1), vel) * TMW_MU(cc);
        o = (e & 1) ? a0 - at : a0 + at;
      }
      L[out + kr] = o;
    }
  }
}
TM_DEV void tmw_jmul(WCtx &c, const WLayout &K, int v, int out) {
  tmw_jmul_stage1(c, K, v);
  TMW_SYNC();
  tmw_jmul_stage2(c, K, v, out);
  TMW_SYNC();
}
// out = J^T f where f_r = active ? -D_r Jaref_r : 0 is formed on the fly (efc_force is never stored)
TM_DEV void tmw_jt_force(WCtx &c, const WLayout &K, int out) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  // (branch-free on purpose: written with `if (penetrating) { .. if (ja < 0) .. }` the compiler emitted a chain of ten dependent LDS round
  // trips, each behind its own exec-mask branch and lgkmcnt(0) wait; here every load of a contact is issued up front — inactive slots read
  // row 0 / their stale frame and select zero at the end)
  // No penetrating contact at all (wave-uniform; ~ 2 of 3 substeps): every wrench and every subset sum is zero — the three passes over the
  // contacts and, below, the dofs' wrench products (which would add exact zeros) are skipped
  const bool anycon = c.nact > c.nla;
  if (anycon) {
  TMW_FOR {
    for (int cc = lane; cc < K.ncon; cc += 64) {
      const bool on = L[K.l_con_dist + cc] < 0.f;
      const int r0b = TMW_CCROW(K)[cc], r0 = on ? r0b : 0;      // first of the contact's four compact rows (255: none)
      float ja[4], D[4], f[4];
#pragma unroll
      for (int e = 0; e < 4; e++) { ja[e] = L[K.l_Jaref + r0 + e]; D[e] = L[K.l_efc_D + TMW_DIDX(r0 + e)]; }
      const float mu = TMW_MU(cc);
      const float *off = L + K.l_con_off + cc * 3;
      const float o0 = off[0], o1 = off[1], o2 = off[2];
      float fr[9];
      tmw_get_con_frame(L, K, cc, fr);
#pragma unroll
      for (int e = 0; e < 4; e++) f[e] = ja[e] < 0.f ? -D[e] * ja[e] : 0.f;
      float c0 = f[0] + f[1] + f[2] + f[3], c1 = mu * (f[0] - f[1]), c2 = mu * (f[2] - f[3]);
      tm_cross(fr + 6, fr, fr + 3);
      float F[3], T[3];
      const float ov[3] = {o0, o1, o2};
      for (int k = 0; k < 3; k++) F[k] = c0 * fr[k] + c1 * fr[3 + k] + c2 * fr[6 + k];
      tm_cross(T, ov, F);
      for (int k = 0; k < 3; k++) { L[K.l_wr + cc * 6 + k] = on ? T[k] : 0.f; L[K.l_wr + cc * 6 + 3 + k] = on ? F[k] : 0.f; }
    }
  }
  TMW_SYNC();
  // wrench of every SUBSET of paw groups that some dof feels (DModel::wsub_*): lane = (subset, component), all contacts
  // streamed past a per-lane contact bit mask; the sums replace the contact wrenches in l_wr
  TMW_REG(float, W0); TMW_REG(float, W1);
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int idx = lane + 64 * slot;
      float sacc = 0.f;
      if (idx < m.n_wsub * 6) {
        int su = idx / 6, k = idx - su * 6;
        // (contact slots 32.. exist only for ncon > 32: their mask word is fetched on demand instead of occupying a register for the whole kernel)
        unsigned m0 = slot == 0 && m.n_wsub * 6 <= 64 ? c.kmask[TMW_LI] : m.wsub_cmask[su][0], m1 = K.ncon > 32 ? m.wsub_cmask[su][1] : 0u;
        // only the ACTIVE contacts carry a wrench: walk their compact rows (four per contact) back to the contact ids
        const unsigned char *rm = TMW_ROWMAP(K);
        for (int kr = c.nla; kr < c.nact; kr += 4) {
          int cc = (rm[kr] - K.nlim) >> 2;
          float w = L[K.l_wr + cc * 6 + k];
          bool in = ((cc < 32 ? m0 >> cc : m1 >> (cc - 32)) & 1u) != 0u;
          sacc += in ? w : 0.f;
        }
      }
      (slot ? W1 : W0)[TMW_LI] = sacc;
      if (m.n_wsub * 6 <= 64) break;
    }
  }
  TMW_SYNC();
  TMW_FOR {
    if (lane < m.n_wsub * 6) L[K.l_wr + lane] = W0[TMW_LI];
    if (lane + 64 < m.n_wsub * 6) L[K.l_wr + lane + 64] = W1[TMW_LI];
  }
  TMW_SYNC();
  }
  TMW_FOR {      // both dof slots level by level (TMW_I1): limit-sign bytes, then the rows' Jaref / D, then the wrench products
    const bool ok1 = TMW_OK1(K);
    const int ii[2] = {lane, TMW_I1(K)};
    int sv[2], su[2], kr[2]; float s[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int i = ii[t];
      // (lean: the limit row of dof i is i - 6 and the wrench subset rides in the dof's packed register word — model_host.h: rodent_chains_match)
      const int lr = K.lean ? (i >= 6 ? i - 6 : -1) : TMW_LIMROW1(TMW_W0(i)) - 1;
      su[t] = K.lean ? TMW_TP_WSUB1(TMW_TP(lane + 64 * t)) - 1 : TMW_WSUB1(TMW_W1(i)) - 1;
      const int svb = TMW_LIMSIGN(K)[lr >= 0 ? lr : 0];      // lim_sign packs sign * (compact row + 1) of a violated limit, 0 otherwise (branch-free: see above)
      sv[t] = lr >= 0 ? svb : 0;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) kr[t] = sv[t] != 0 ? (sv[t] < 0 ? -sv[t] : sv[t]) - 1 : 0;
    const float ja0 = L[K.l_Jaref + kr[0]], D0 = L[K.l_efc_D + TMW_DIDX(kr[0])], ja1 = L[K.l_Jaref + kr[1]], D1 = L[K.l_efc_D + TMW_DIDX(kr[1])];
    { const float fl = -D0 * ja0; s[0] = (sv[0] != 0 && ja0 < 0.f) ? (sv[0] > 0 ? fl : -fl) : 0.f; }
    { const float fl = -D1 * ja1; s[1] = (sv[1] != 0 && ja1 < 0.f) ? (sv[1] > 0 ? fl : -fl) : 0.f; }
    if (anycon) {
      float cd[2][6], w[2][6];
#pragma unroll
      for (int t = 0; t < 2; t++) { const int sw = su[t] >= 0 ? su[t] : 0; for (int k = 0; k < 6; k++) { cd[t][k] = L[K.l_cdof + ii[t] * 6 + k]; w[t][k] = L[K.l_wr + sw * 6 + k]; } }
#pragma unroll
      for (int t = 0; t < 2; t++) { float a = s[t]; for (int k = 0; k < 6; k++) a += cd[t][k] * w[t][k]; s[t] = su[t] >= 0 ? a : s[t]; }
    }
    L[out + lane] = s[0]; if (ok1) L[out + ii[1]] = s[1];
  }
  TMW_SYNC();
}

// y = M x and out = J x together: the two products are independent, so their LDS round trips share the two barriers of
// M x instead of adding two more (chain layout only; the generic path runs them back to back)
TM_DEV void tmw_mul_m_jmul(WCtx &c, const WLayout &K, int x, int y, int out) {
  float *L = c.L; TMW_LANE_DECL
  if (!K.chains) { tmw_mul_m(c, K, x, y); tmw_jmul(c, K, x, out); return; }
  TMW_REG(float, z0); TMW_REG(float, z1);
  TMW_TICK2(15);
  tmw_jmul_stage1(c, K, x);
  tmw_rowpart_chains(c, K, K.l_M, x, true, z0, z1);
  TMW_SCHED_FENCE();
  tmw_colpart_chains<false>(c, K, K.l_M, x, y);     // ends with a barrier: l_sv is complete as well
  tmw_jmul_stage2(c, K, x, out);
  TMW_FOR { L[y + lane] += z0[TMW_LI]; if (lane + 64 < K.nv) L[y + lane + 64] += z1[TMW_LI]; }
  TMW_SYNC();
  TMW_TICK2(24);
}

// ------------------------------------------------------------------------------------------ make_constraint
// Only ACTIVE rows enter the solver: a limit row exists iff the limit is violated, the four pyramid rows of a contact iff it
// penetrates (dist < 0).  MJX keeps all 187 rows, but an inactive row has a zero Jacobian row and aref <= 0, so Jaref >= 0
// for every qacc and the row contributes exactly 0 to the cost, its gradient and every line-search sum.  The active rows are
// numbered consecutively in the original order (limits, then contacts); l_rowmap maps back, lim_sign / l_ccrow map forward.
// Typical counts (a few limits + 2..8 contacts) fit ONE 64-row slot instead of three.
TM_DEV void tmw_make_constraint(WCtx &c, const WLayout &K) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_REG(int, f0); TMW_REG(int, f1); TMW_REG(int, fc); TMW_REG(int, x0); TMW_REG(int, x1); TMW_REG(int, xc);
  TMW_REG(float, s0); TMW_REG(float, s1); TMW_REG(int, d0); TMW_REG(int, d1);
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int l = lane + 64 * slot;
      float sg = 0.f;
      (slot ? d1 : d0)[TMW_LI] = 0;
      if (l < K.nlim) {
        int j = m.lim_jnt[l];
        (slot ? d1 : d0)[TMW_LI] = m.jnt_dofadr[j];
        float q = L[K.l_qpos + m.jnt_qposadr[j]], dmin = q - m.jnt_range[j][0], dmax = m.jnt_range[j][1] - q;
        float pos = fminf(dmin, dmax) - m.jnt_margin[j];
        sg = pos < 0.f ? (dmin < dmax ? 1.f : -1.f) : 0.f;
      }
      (slot ? s1 : s0)[TMW_LI] = sg; (slot ? f1 : f0)[TMW_LI] = sg != 0.f;
    }
    fc[TMW_LI] = lane < K.ncon && L[K.l_con_dist + (lane < K.ncon ? lane : 0)] < 0.f;
  }
  const int n0 = tmw_prefix(f0, x0), n1 = tmw_prefix(f1, x1), ncl = tmw_prefix(fc, xc), nla = n0 + n1;
  c.nact = nla + 4 * ncl; c.nla = nla;
  TMW_FOR {
    unsigned char *rm = TMW_ROWMAP(K), *cr = TMW_CCROW(K);
    // map byte of a limit row: its dof | (violated on the upper side) << 7 — all that J and J^T need; of a contact row: nlim + 4 cc + e
    if (lane < K.nlim) { TMW_LIMSIGN(K)[lane] = (signed char)(s0[TMW_LI] * (float)(x0[TMW_LI] + 1)); if (f0[TMW_LI]) rm[x0[TMW_LI]] = (unsigned char)(d0[TMW_LI] | (s0[TMW_LI] < 0.f ? 0x80 : 0)); }
    if (lane + 64 < K.nlim) { TMW_LIMSIGN(K)[lane + 64] = (signed char)(s1[TMW_LI] * (float)(n0 + x1[TMW_LI] + 1)); if (f1[TMW_LI]) rm[n0 + x1[TMW_LI]] = (unsigned char)(d1[TMW_LI] | (s1[TMW_LI] < 0.f ? 0x80 : 0)); }
    if (lane < K.ncon) {
      int r0 = nla + 4 * xc[TMW_LI];
      cr[lane] = fc[TMW_LI] ? (unsigned char)r0 : (unsigned char)255;
      if (fc[TMW_LI]) for (int e = 0; e < 4; e++) rm[r0 + e] = (unsigned char)(K.nlim + 4 * lane + e);
    }
  }
  TMW_SYNC();
  tmw_jmul(c, K, K.l_qvel, K.l_jv);
  TMW_FOR {
    const unsigned char *rm = TMW_ROWMAP(K);
    for (int kr = lane; kr < c.nact; kr += 64) {
      int r = rm[kr];
      float k, b, imp, pos, iw;
      float solref[2], solimp[5];        // ONE impedance evaluation for limit and contact rows (both kinds share a wave)
      if (kr < c.nla) {
        TMW_LOCAL(rec, 12, m.dof_lim[r & 0x7f]);      // the dof's limit record: one model read behind the row-map byte (was dof -> joint -> fields)
        float q = L[K.l_qpos + tm_f2i(rec[0])], dmin = q - rec[1], dmax = rec[2] - q;
        pos = fminf(dmin, dmax) - rec[3];
        for (int t = 0; t < 2; t++) solref[t] = rec[4 + t];
        for (int t = 0; t < 5; t++) solimp[t] = rec[6 + t];
        iw = rec[11];
      } else {
        int cc = (r - K.nlim) >> 2;
        pos = L[K.l_con_dist + cc];
        for (int t = 0; t < 2; t++) solref[t] = m.con_solref[cc][t];
        for (int t = 0; t < 5; t++) solimp[t] = m.con_solimp[cc][t];
        iw = m.con_invweight[cc];
      }
      tm_kbi(m.timestep, solref, solimp, pos, k, b, imp);
      float Rr = fmaxf(iw * (1.f - imp) / imp, TM_MINVAL);
      L[K.l_efc_D + TMW_DIDX(kr)] = 1.f / Rr;        // (lean: the four rows of a contact write the same word with the same value)
      L[K.l_efc_aref + kr] = -b * L[K.l_jv + kr] - k * imp * pos;
    }
  }
  TMW_SYNC();
  if (c.dump) {   // tests only: efc_D / efc_aref of ALL rows in the original order (inactive rows: J v = 0)
    TMW_FOR {
      for (int r = lane; r < K.nefc; r += 64) {
        float k, b, imp, pos, iw; int kr = -1;
        if (r < K.nlim) {
          int j = m.lim_jnt[r];
          float q = L[K.l_qpos + m.jnt_qposadr[j]], dmin = q - m.jnt_range[j][0], dmax = m.jnt_range[j][1] - q;
          pos = fminf(dmin, dmax) - m.jnt_margin[j];
          { TMW_LOCAL(sr_, 2, m.jnt_solref[j]); TMW_LOCAL(si_, 5, m.jnt_solimp[j]); tm_kbi(m.timestep, sr_, si_, pos, k, b, imp); }
          iw = m.dof_invweight0[m.jnt_dofadr[j]];
          float sv = (float)TMW_LIMSIGN(K)[r];
          if (sv != 0.f) kr = (int)fabsf(sv) - 1;
        } else {
          int cc = (r - K.nlim) >> 2;
          pos = L[K.l_con_dist + cc];
          { TMW_LOCAL(sr_, 2, m.con_solref[cc]); TMW_LOCAL(si_, 5, m.con_solimp[cc]); tm_kbi(m.timestep, sr_, si_, pos, k, b, imp); }
          iw = m.con_invweight[cc];
          if (TMW_CCROW(K)[cc] != 255) kr = TMW_CCROW(K)[cc] + ((r - K.nlim) & 3);
        }
        float Rr = fmaxf(iw * (1.f - imp) / imp, TM_MINVAL);
        c.dump[(size_t)(m.w_efc_D + r) * (size_t)c.n + (size_t)c.e] = 1.f / Rr;
        // (l_jv aliases l_efc_aref: the active rows' J v has already been folded into aref above)
        c.dump[(size_t)(m.w_efc_aref + r) * (size_t)c.n + (size_t)c.e] = kr >= 0 ? L[K.l_efc_aref + kr] : -k * imp * pos;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ CG solver
TM_DEV float tmw_dot(WCtx &c, const WLayout &K, int a, int b) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_REG(float, p);
  TMW_FOR { float s = 0.f; for (int i = lane; i < K.nv; i += 64) s += L[a + i] * L[b + i]; p[TMW_LI] = s; }
  return tmw_sum(p);
}
// Jaref <- J q - aref ; Ma <- M q ; returns cost and gauss for qacc vector `q`
// `jonly`: q IS qacc_smooth = M^-1 qfrc_smooth: the Gauss term is exactly zero (q - qacc_smooth = 0 in MJX's own arithmetic as
// well) and M q is not needed by the caller (tmw_solve_cg) — no product with M
// `have_ma`: l_Ma already holds M q (chain layout: taken before M was factorised in place) — J q only, Gauss term from l_Ma
TM_DEV float tmw_eval_cost(WCtx &c, const WLayout &K, int q, float &gauss, bool jonly = false, bool have_ma = false) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_TICK2(15);
  if (jonly || have_ma) tmw_jmul(c, K, q, K.l_Jaref);
  else tmw_mul_m_jmul(c, K, q, K.l_Ma, K.l_Jaref);
  TMW_TICK2(19);
  TMW_REG(float, pc); TMW_REG(float, pg);
  TMW_FOR {
    float sc = 0.f, sg = 0.f;
    for (int e = lane; e < c.nact; e += 64) {
      float ja = L[K.l_Jaref + e] - L[K.l_efc_aref + e];
      L[K.l_Jaref + e] = ja;
      if (ja < 0.f) sc += L[K.l_efc_D + TMW_DIDX(e)] * ja * ja;
    }
    // (lean: the only evaluation with a Gauss term is the warm start's, q = l_qacc: iterate and M q come from their registers)
    if (!jonly) {
      const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
      const float qs0 = L[K.l_qacc_smooth + lane], qs1 = L[K.l_qacc_smooth + i1];
      const float q0 = TMW_QAREG(K) ? c.qa0[TMW_LI] : L[q + lane], q1 = TMW_QAREG(K) ? c.qa1[TMW_LI] : L[q + i1];
      sg += (TMW_MA(lane) - TMW_QFS(lane)) * (q0 - qs0);
      sg += (ok1 ? TMW_MA1(K, i1) - (K.lean ? c.qfs1[TMW_LI] : L[K.l_qfrc_smooth + i1]) : 0.f) * (q1 - qs1);
    }
    pc[TMW_LI] = sc; pg[TMW_LI] = sg;
  }
  TMW_SYNC();
  gauss = jonly ? 0.f : 0.5f * tmw_sum(pg);
  return 0.5f * tmw_sum(pc) + gauss;
}
// ---- The CG iteration runs in the coordinates y = L qacc of the factorisation M = L^T D L (N = L^-1 sits in l_LD):
//   Gauss term      1/2 (y - y_s)^T D (y - y_s)            (diagonal: no product with M inside the iteration)
//   preconditioner  M^-1 = N D^-1 N^T  ->  D^-1            (diagonal)
//   search_q = N s  (root -> leaf half of the M^-1 product),  w = D^-1 N^T grad_q  (leaf -> root half)
// so one iteration costs ONE M^-1 product split into its two halves instead of M x + M^-1 x.  In exact arithmetic the
// iterates are those of MJX's solver (same Polak-Ribiere directions: g.M^-1 g = ghat.D^-1 ghat, search.M search = s.D s, ...);
// the state is self-consistent in fp32 (ut = y - y_s is advanced by the step actually taken).  LDS vectors of the y-space
// quantities (names of the q-space vectors they replace): l_Ma = ut = y - y_s, l_Mgrad = w = D^-1 ghat, registers wp* = previous w,
// l_mv = s, registers dg* = D;  l_search = search_q, l_qacc, l_qfrc_constraint stay in q-space.
// x -> out = D^-1 N^T x   (leaf -> root; out may alias x)
TM_DEV void tmw_solve_up(WCtx &c, const WLayout &K, int x, int out) {
  float *L = c.L; TMW_LANE_DECL
  if (K.chains) { tmw_colpart_chains<true>(c, K, K.l_LD, x, out); return; }
  TMW_REG(float, z0); TMW_REG(float, z1);
  TMW_FOR {
    for (int slot = 0; slot < 2; slot++) {
      int i = lane + 64 * slot;
      if (i >= K.nv) continue;
      (slot ? z1 : z0)[TMW_LI] = (L[x + i] + tmw_col_dot(L, K, K.l_LD, x, i)) * L[K.l_Dinv + i];
    }
  }
  TMW_SYNC();
  TMW_FOR { for (int slot = 0; slot < 2; slot++) { int i = lane + 64 * slot; if (i < K.nv) L[out + i] = (slot ? z1 : z0)[TMW_LI]; } }
  TMW_SYNC();
}
// out = N x   (root -> leaf; out may NOT alias x)
TM_DEV void tmw_solve_down(WCtx &c, const WLayout &K, int x, int out) {
  float *L = c.L; TMW_LANE_DECL
  if (K.chains) {
    TMW_REG(float, z0); TMW_REG(float, z1);
    tmw_rowpart_chains(c, K, K.l_LD, x, false, z0, z1);
    TMW_FOR { L[out + lane] = z0[TMW_LI] + L[x + lane]; if (lane + 64 < K.nv) L[out + lane + 64] = z1[TMW_LI] + L[x + lane + 64]; }
    TMW_SYNC();
    return;
  }
  TMW_FOR { for (int i = lane; i < K.nv; i += 64) L[out + i] = L[x + i] + tmw_row_dot(L, K, K.l_LD, x, i, false); }
  TMW_SYNC();
}
// qfrc_constraint = J^T f at the current Jaref;  w = ut - D^-1 N^T qfrc_constraint;  returns gn = w.D w (= grad.M^-1 grad) and,
// through `num`, w.D (w - w_prev) with w_prev = the wp registers (Polak-Ribiere numerator); `improvement < last_if`: qfrc_constraint only, returns -1
TM_DEV float tmw_update_gradient(WCtx &c, const WLayout &K, float &num, float improvement = 0.f, float last_if = -INFINITY) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_TICK2(15);
  tmw_jt_force(c, K, K.l_qfrc_constraint);
  TMW_TICK2(16);
  if (improvement < last_if) return -1.f;      // the solve's last pass (tmw_solve_cg): qfrc_constraint only
  tmw_solve_up(c, K, K.l_qfrc_constraint, K.l_Mgrad);
  TMW_TICK2(26);
  TMW_REG(float, pa); TMW_REG(float, pb);
  TMW_FOR {
    float sa = 0.f, sb = 0.f;
    const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
    const float g0 = L[K.l_Mgrad + lane], g1 = L[K.l_Mgrad + i1];
    const float w0 = TMW_MA(lane) - g0, dw0 = c.dg0[TMW_LI] * w0, w1 = TMW_MA1(K, i1) - g1, dw1 = c.dg1[TMW_LI] * w1;
    L[K.l_Mgrad + lane] = w0; if (ok1) L[K.l_Mgrad + i1] = w1;
    sa += dw0 * w0; sb += dw0 * (w0 - c.wp0[TMW_LI]);
    sa += (ok1 ? dw1 : 0.f) * w1; sb += (ok1 ? dw1 : 0.f) * (w1 - c.wp1[TMW_LI]);
    pa[TMW_LI] = sa; pb[TMW_LI] = sb;
  }
  TMW_SYNC();
  num = tmw_sum(pb);
  float gn = tmw_sum(pa);
  TMW_TICK2(17);
  return gn;
}
// A value rounded to float32 where it stands: the compiler may not fuse the product behind it into a following add.  The line search
// needs this for the derivative  d0 = 2 alpha q2 + q1:  at the Newton point alpha = -q1 / (2 q2) the ROUNDED product cancels q1 exactly
// about half of the time, d0 == 0 is a fixed point of the bracket update and the search stops (MJX's behaviour, and the oracle's);
// fused (hipcc contracts a * b + c by default) d0 is the exact, almost never zero, rounding residual of the division and the search keeps
// refining a converged bracket: +28 % line-search iterations for the same result (tests/diagnostics/ls_iterations.py).
#if defined(TMW_FUSED_D0)      // the diagnostic's "before" arm
TM_DEV float tmw_round(float p) { return p; }
#elif defined(TM_HOST_EMU)
TM_DEV float tmw_round(float p) { volatile float q = p; return q; }
#else
TM_DEV float tmw_round(float p) { asm volatile("" : "+v"(p)); return p; }
#endif
struct TmwLS { float alpha, cost, d0, d1; };
// The constraint rows of a line search live in registers: per row e = lane + 64 s the three quadratic coefficients
// t0 = D ja^2 / 2, t1 = D ja jv, t2 = D jv^2 / 2 and (ja, jv) for the activity test ja + alpha jv < 0.
#define TMW_LS_SLOTS 4        // up to 256 constraint rows
struct TmwLSRows { float ja[TMW_NL][TMW_LS_SLOTS], jv[TMW_NL][TMW_LS_SLOTS], D[TMW_NL][TMW_LS_SLOTS], t0[TMW_NL][TMW_LS_SLOTS], t1[TMW_NL][TMW_LS_SLOTS], t2[TMW_NL][TMW_LS_SLOTS]; };
// three row sums side by side, IN PLACE: afterwards lane 16 r + 15 of each array holds the sum over row r.  Written step-interleaved: a
// DPP operand needs two wait states after the instruction that wrote it, and the three independent chains fill each other's slots (one
// after the other they were 8 s_nop per evaluation)
#ifdef TM_HOST_EMU
TM_DEV void tmw_rowscan16x3(float *a, float *b, float *d) {
  for (int r = 0; r < 4; r++) { float sa = 0.f, sb = 0.f, sd = 0.f; for (int l = 0; l < 16; l++) { sa += a[16 * r + l]; sb += b[16 * r + l]; sd += d[16 * r + l]; }
    a[16 * r + 15] = sa; b[16 * r + 15] = sb; d[16 * r + 15] = sd; }
}
#else
TM_DEV void tmw_rowscan16x3(float *a, float *b, float *d) {
  float x = a[0], y = b[0], z = d[0];
  x = tmw_dpp_add<0x111, 0xf, 0xf>(x); y = tmw_dpp_add<0x111, 0xf, 0xf>(y); z = tmw_dpp_add<0x111, 0xf, 0xf>(z);
  x = tmw_dpp_add<0x112, 0xf, 0xf>(x); y = tmw_dpp_add<0x112, 0xf, 0xf>(y); z = tmw_dpp_add<0x112, 0xf, 0xf>(z);
  x = tmw_dpp_add<0x114, 0xf, 0xf>(x); y = tmw_dpp_add<0x114, 0xf, 0xf>(y); z = tmw_dpp_add<0x114, 0xf, 0xf>(z);
  x = tmw_dpp_add<0x118, 0xf, 0xf>(x); y = tmw_dpp_add<0x118, 0xf, 0xf>(y); z = tmw_dpp_add<0x118, 0xf, 0xf>(z);
  a[0] = x; b[0] = y; d[0] = z;
}
#endif
// At most 16 active rows (the usual case): the rows are replicated in lanes 0-15, 16-31 and 32-47 (tmw_linesearch) and the three
// candidates of an iteration are evaluated side by side, one per 16-lane row — 3 masked sums and 3 row reductions instead of 9 + 9.
// Cost and derivatives are then formed IN the last lane of each row, for the row's own alpha, by one instruction stream for all three
// candidates; only those three numbers per candidate are broadcast (broadcasting the sums and repeating the arithmetic per candidate on
// wave-uniform values was 36 instead of 12 vector instructions per evaluation).
template <int NP>
TM_DEV void tmw_ls_points16(WCtx &c, const TmwLSRows &R, const float *a, float g0, float g1, float g2, TmwLS *out) {
  TMW_LANE_DECL
  TMW_REG(float, q0); TMW_REG(float, q1); TMW_REG(float, q2); TMW_REG(float, av);
  TMW_FOR {
    float al = a[0];
    if (NP > 1) al = TMW_MASK(TMW_M_LT(16)) ? a[0] : (TMW_MASK(TMW_M_LT(32)) ? a[1] : a[2]);
    bool act = R.ja[TMW_LI][0] + al * R.jv[TMW_LI][0] < 0.f;
    q0[TMW_LI] = act ? R.t0[TMW_LI][0] : 0.f; q1[TMW_LI] = act ? R.t1[TMW_LI][0] : 0.f; q2[TMW_LI] = act ? R.t2[TMW_LI][0] : 0.f;
    av[TMW_LI] = al;
  }
  tmw_rowscan16x3(q0, q1, q2);
  TMW_REG(float, pc); TMW_REG(float, pd0); TMW_REG(float, pd1);
  TMW_FOR {
    float r0 = g0 + q0[TMW_LI], r1 = g1 + q1[TMW_LI], r2 = g2 + q2[TMW_LI], al = av[TMW_LI];
    pc[TMW_LI] = al * al * r2 + al * r1 + r0;
    pd0[TMW_LI] = tmw_round(2.f * al * r2) + r1;
    pd1[TMW_LI] = 2.f * r2 + (r2 == 0.f ? TM_MINVAL : 0.f);
  }
#pragma unroll
  for (int p = 0; p < NP; p++) {
    out[p].alpha = a[p];
    out[p].cost = tmw_readlane(pc, 16 * p + 15); out[p].d0 = tmw_readlane(pd0, 16 * p + 15); out[p].d1 = tmw_readlane(pd1, 16 * p + 15);
  }
}
// NP line-search points evaluated together (alphas a[0..NP-1]): 3 NP partial sums over the rows, reduced together
template <int NP, int WIDTH>
TM_DEV void tmw_ls_points(WCtx &c, const WLayout &K, const TmwLSRows &R, const float *a, float g0, float g1, float g2, TmwLS *out) {
  if (WIDTH == 16) { tmw_ls_points16<NP>(c, R, a, g0, g1, g2, out); return; }
  TMW_LANE_DECL
  float q[3 * NP][TMW_NL];
  const int nslot = (c.nact + 63) / 64;
  TMW_FOR {
    float s[3 * NP];
    for (int k = 0; k < 3 * NP; k++) s[k] = 0.f;
#pragma unroll
    for (int sl = 0; sl < TMW_LS_SLOTS; sl++) {
      if (sl >= nslot) break;
      float ja = R.ja[TMW_LI][sl], jv = R.jv[TMW_LI][sl];
#pragma unroll
      for (int p = 0; p < NP; p++) {
        bool act = ja + a[p] * jv < 0.f;
        s[3 * p] += act ? R.t0[TMW_LI][sl] : 0.f; s[3 * p + 1] += act ? R.t1[TMW_LI][sl] : 0.f; s[3 * p + 2] += act ? R.t2[TMW_LI][sl] : 0.f;
      }
    }
    for (int k = 0; k < 3 * NP; k++) q[k][TMW_LI] = s[k];
  }
  float qq[3 * NP];
#pragma unroll
  for (int k = 0; k < 3 * NP; k++) qq[k] = tmw_sum_w<WIDTH>(q[k]);
#pragma unroll
  for (int p = 0; p < NP; p++) {
    float q0 = g0 + qq[3 * p], q1 = g1 + qq[3 * p + 1], q2 = g2 + qq[3 * p + 2], al = a[p];
    out[p].alpha = al;
    out[p].cost = al * al * q2 + al * q1 + q0;
    out[p].d0 = tmw_round(2.f * al * q2) + q1;
    out[p].d1 = 2.f * q2 + (q2 == 0.f ? TM_MINVAL : 0.f);
  }
}
// the iteration proper (MJX _linesearch: Newton steps from both ends of the bracket + the midpoint), for a compile-time width of
// the wave reductions: WIDTH = 16 / 32 when all active rows sit in the first 16 / 32 lanes (the usual case), else 64
template <int WIDTH>
TM_DEV float tmw_ls_core(WCtx &c, const WLayout &K, const TmwLSRows &R, float g0, float g1, float g2, float gtol) {
  TmwModel &m = *c.mp; float *L = c.L;
  TmwLS pt[3];
  float al[3] = {0.f, 0.f, 0.f};
  tmw_ls_points<1, WIDTH>(c, K, R, al, g0, g1, g2, pt);
  TmwLS p0 = pt[0];
  al[0] = p0.alpha - p0.d0 * tmw_rcp(p0.d1);     // d1 = 2 q2 > 0: hardware reciprocal + one Newton step instead of the IEEE division sequence
  tmw_ls_points<1, WIDTH>(c, K, R, al, g0, g1, g2, pt);
  TmwLS lo0 = pt[0];
  TMW_TICK2(35);
#ifndef TMW_LS_NO_SHORTCUT
  // d0 == 0 EXACTLY at the Newton point — the usual outcome when no row changes sides on the way (tmw_round).  What the loop below then
  // does is determined: its first iteration moves both ends of the bracket onto this point (whatever the midpoint evaluates to), the
  // second finds nothing to swap.  Those six evaluations are skipped; the step returned and the iteration count reported are the loop's.
  if (lo0.d0 == 0.f) {
    const bool done0 = (p0.d0 < 0.f && p0.d0 > -gtol) || (p0.d0 > 0.f && p0.d0 < gtol);
    const int cnt = done0 ? 0 : (p0.d0 == 0.f ? 1 : 2);
    TMW_STATS(K) += (float)(cnt < m.ls_iterations ? cnt : m.ls_iterations);
    TMW_COUNT(38, 1000);
    return lo0.cost < p0.cost ? lo0.alpha : 0.f;
  }
#endif
  bool lesser = lo0.d0 < p0.d0;
  TmwLS hi = lesser ? p0 : lo0, lo = lesser ? lo0 : p0;
  bool swap = true;
  int it = 0;
  for (; it < m.ls_iterations; it++) {
    bool done = !swap || ((lo.d0 < 0.f) && (lo.d0 > -gtol)) || ((hi.d0 > 0.f) && (hi.d0 < gtol));
    if (done) break;
#ifndef TMW_LS_NO_SHORTCUT
    // the bracket has collapsed onto a point with d0 == 0: all three candidates of this iteration ARE that point, nothing swaps
    if (lo.alpha == hi.alpha && lo.d0 == 0.f && hi.d0 == 0.f) { it++; break; }
#endif
    al[0] = lo.alpha - lo.d0 * tmw_rcp(lo.d1); al[1] = hi.alpha - hi.d0 * tmw_rcp(hi.d1); al[2] = 0.5f * (lo.alpha + hi.alpha);
    tmw_ls_points<3, WIDTH>(c, K, R, al, g0, g1, g2, pt);
    TmwLS lo_next = pt[0], hi_next = pt[1], mid = pt[2];
    bool s1 = (lo.d0 > 0.f) || (lo.d0 < lo_next.d0);
    if (s1) lo = lo_next;
    bool s2 = (mid.d0 < 0.f) && (lo.d0 < mid.d0);
    if (s2) lo = mid;
    bool s3 = (hi.d0 < 0.f) || (hi.d0 > hi_next.d0);
    if (s3) hi = hi_next;
    bool s4 = (mid.d0 > 0.f) && (hi.d0 > mid.d0);
    if (s4) hi = mid;
    swap = s1 || s2 || s3 || s4;
  }
  TMW_STATS(K) += (float)it;
  TMW_COUNT(37, 1000 * it);
  bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  return improved ? alpha : 0.f;
}
// returns the cost of the new iterate and its Gauss term through `gauss` (what _update_constraint recomputes after the step:
// 1/2 sum D ja^2 over the active rows + 1/2 ut.D ut, from the rows and vectors already in registers)
TM_DEV float tmw_linesearch(WCtx &c, const WLayout &K, float &gauss) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  float scale = m.meaninertia * (float)(K.nv > 1 ? K.nv : 1);
  tmw_solve_down(c, K, K.l_mv, K.l_search);            // search_q = N s
  TMW_TICK2(32);
  tmw_jmul(c, K, K.l_search, K.l_jv);
  TMW_TICK2(33);
  // |search_q|^2, s.D ut (= search.Ma - search.qfrc_smooth) and s.D s (= search.M search) in one pass, three reductions side by side
  TMW_REG(float, p0); TMW_REG(float, p1); TMW_REG(float, p2);
  TMW_FOR {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
    const float sq0 = L[K.l_search + lane], sq1 = L[K.l_search + i1], m0 = L[K.l_mv + lane], m1 = L[K.l_mv + i1];
    const float ds0 = c.dg0[TMW_LI] * m0, ds1 = ok1 ? c.dg1[TMW_LI] * m1 : 0.f;
    a0 += sq0 * sq0; a1 += ds0 * TMW_MA(lane); a2 += ds0 * m0;
    a0 += (ok1 ? sq1 : 0.f) * sq1; a1 += ds1 * TMW_MA1(K, i1); a2 += ds1 * m1;
    p0[TMW_LI] = a0; p1[TMW_LI] = a1; p2[TMW_LI] = a2;
  }
  float smag = sqrtf(tmw_sum(p0)) * scale;
  float gtol = m.tolerance * m.ls_tolerance * smag;
  float g0 = gauss, g1 = tmw_sum(p1), g2 = 0.5f * tmw_sum(p2);
  TMW_TICK2(34);
  TmwLSRows R;
  const bool w16 = c.nact <= 16;
  TMW_FOR {
#pragma unroll
    for (int sl = 0; sl < TMW_LS_SLOTS; sl++) {
      // (wave-uniform: no row in this slot — slots 1 .. 3 in four solves of five: the padding row's constants without its three LDS reads)
      if (sl > 0 && (w16 || 64 * sl >= c.nact)) {
        R.ja[TMW_LI][sl] = 1.f; R.jv[TMW_LI][sl] = 0.f; R.D[TMW_LI][sl] = 0.f; R.t0[TMW_LI][sl] = 0.f; R.t1[TMW_LI][sl] = 0.f; R.t2[TMW_LI][sl] = 0.f;
        continue;
      }
      // nact <= 16: the rows sit in lanes 0-15 and are replicated in lanes 16-31 and 32-47 (tmw_ls_points16)
      int e = w16 ? (lane & 15) : lane + 64 * sl;
      bool ok = w16 ? (sl == 0 && lane < 48 && e < c.nact) : e < c.nact;
      float ja = ok ? L[K.l_Jaref + (ok ? e : 0)] : 1.f, jv = ok ? L[K.l_jv + (ok ? e : 0)] : 0.f, D = ok ? L[K.l_efc_D + (ok ? TMW_DIDX(e) : 0)] : 0.f;   // padding rows: never active
      R.ja[TMW_LI][sl] = ja; R.jv[TMW_LI][sl] = jv; R.D[TMW_LI][sl] = D;
      R.t0[TMW_LI][sl] = 0.5f * ja * ja * D; R.t1[TMW_LI][sl] = jv * ja * D; R.t2[TMW_LI][sl] = 0.5f * jv * jv * D;
    }
  }
  TMW_TICK2(13);
  // one uniform branch picks the reduction width for the whole search: the active rows fill lanes 0 .. nact-1 of slot 0
  float ia;
  if (c.nact <= 16) ia = tmw_ls_core<16>(c, K, R, g0, g1, g2, gtol);
  else if (c.nact <= 32) ia = tmw_ls_core<32>(c, K, R, g0, g1, g2, gtol);
  else ia = tmw_ls_core<64>(c, K, R, g0, g1, g2, gtol);
  TMW_TICK2(14);
  TMW_REG(float, pc); TMW_REG(float, pg);
  TMW_FOR {
    float sc = 0.f, sg = 0.f;
    {
      const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
      const float sq0 = L[K.l_search + lane], sq1 = L[K.l_search + i1], m0 = L[K.l_mv + lane], m1 = L[K.l_mv + i1];
      const float qa0 = TMW_QA(lane) + sq0 * ia, ut0 = TMW_MA(lane) + m0 * ia;
      const float qa1 = TMW_QA1(K, i1) + sq1 * ia, ut1 = TMW_MA1(K, i1) + m1 * ia;
      TMW_QA_SET(lane, qa0); TMW_MA_SET(lane, ut0);
      if (ok1) { TMW_QA_SET(lane + 64, qa1); TMW_MA_SET(lane + 64, ut1); }
      sg += c.dg0[TMW_LI] * ut0 * ut0;
      sg += (ok1 ? c.dg1[TMW_LI] * ut1 : 0.f) * ut1;
    }
#pragma unroll
    for (int sl = 0; sl < TMW_LS_SLOTS; sl++) {
      int e = lane + 64 * sl;
      if (e < c.nact) { float ja = R.ja[TMW_LI][sl] + R.jv[TMW_LI][sl] * ia; L[K.l_Jaref + e] = ja; if (ja < 0.f) sc += R.D[TMW_LI][sl] * ja * ja; }
    }
    pc[TMW_LI] = sc; pg[TMW_LI] = sg;
  }
  TMW_SYNC();
  gauss = 0.5f * tmw_sum(pg);
  return 0.5f * tmw_sum(pc) + gauss;
}
TM_DEV void tmw_solve_cg(WCtx &c, const WLayout &K) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  float gauss, scale = m.meaninertia * (float)(K.nv > 1 ? K.nv : 1);
#if defined(TMW_PROFILE) && !defined(TM_HOST_EMU)
  if (c.prof && c.lane == 0) { c.prof[10] += (c.nact <= 16) ? 1000 : 0; c.prof[18] += (c.nact > 16 && c.nact <= 32) ? 1000 : 0; }   // histogram of active rows (x1000) in two unused slots
#endif
  TMW_FOR {
    c.dg1[TMW_LI] = 0.f; c.wp1[TMW_LI] = 0.f;
    if (TMW_QAREG(K)) { c.qa1[TMW_LI] = 0.f; c.ma1[TMW_LI] = 0.f; }
    {      // (both slots' global and LDS loads first: the warm start's global round trip once, not once per slot)
      const bool ok1 = TMW_OK1(K); const int i1 = TMW_I1(K);
      const float wq0 = WST(m.s_warm, lane), wq1 = WST(m.s_warm, i1);
      const float ma0 = L[K.l_Ma + lane], ma1 = L[K.l_Ma + i1], di0 = L[K.l_Dinv + lane], di1 = L[K.l_Dinv + i1];
      L[K.l_qacc + lane] = wq0;                                       // (lean: the staging image J qacc reads below)
      if (ok1) L[K.l_qacc + i1] = wq1;
      if (TMW_QAREG(K)) { c.qa0[TMW_LI] = wq0; c.ma0[TMW_LI] = ma0; if (ok1) { c.qa1[TMW_LI] = wq1; c.ma1[TMW_LI] = ma1; } }     // ... and M * warm start, parked in l_Ma = l_Mgrad since tmw_forward took it
      c.dg0[TMW_LI] = 1.f / di0; c.wp0[TMW_LI] = 0.f;
      if (ok1) { c.dg1[TMW_LI] = 1.f / di1; c.wp1[TMW_LI] = 0.f; }
    }
  }
  TMW_SYNC();
  // start from the warm start unless the unconstrained acceleration has the lower cost (MJX evaluates warm, smooth and
  // then the winner again; evaluating smooth FIRST leaves M qacc / Jaref of the warm start — the usual winner — in place)
  float gs, cs = tmw_eval_cost(c, K, K.l_qacc_smooth, gs, true);
  float cw = tmw_eval_cost(c, K, K.l_qacc, gauss, false, K.m_spilled());
  float cost = cw;
  if (cw < cs) {     // ut = y - y_s = D^-1 N^T (M qacc - qfrc_smooth)
    TMW_FOR { for (int i = lane; i < K.nv; i += 64) L[K.l_Ma + i] -= TMW_QFS(i); }      // (lean: on the parked image, still intact)
    TMW_SYNC();
    tmw_solve_up(c, K, K.l_Ma, K.l_Ma);
    if (TMW_QAREG(K)) { TMW_FOR { for (int i = lane; i < K.nv; i += 64) TMW_MA_SET(i, L[K.l_Ma + i]); } }
  } else {
    TMW_FOR { for (int i = lane; i < K.nv; i += 64) { const float qs = L[K.l_qacc_smooth + i]; if (TMW_QAREG(K)) { TMW_QA_SET(i, qs); TMW_MA_SET(i, 0.f); } else { L[K.l_qacc + i] = qs; L[K.l_Ma + i] = 0.f; } } }
    TMW_SYNC();
    cost = tmw_eval_cost(c, K, TMW_QAREG(K) ? K.l_qacc_smooth : K.l_qacc, gauss, true);      // (J q only: the same vector, read where it lies)
  }
  float prev_cost = INFINITY, num;
  float gn = tmw_update_gradient(c, K, num);
  TMW_FOR { const int i1 = TMW_I1(K); const float g0 = L[K.l_Mgrad + lane], g1 = L[K.l_Mgrad + i1]; L[K.l_mv + lane] = -g0; if (TMW_OK1(K)) L[K.l_mv + i1] = -g1; }
  TMW_SYNC();
  TMW_TICK(6);
  TMW_STATS(K) = 0.f;
  int it = 0;
  for (; it < m.iterations; it++) {
    if (m.iterations != 1) {
      // MJX tests |grad_q| / scale; grad_q = L^T ghat is not formed here: its M^-1-norm (gn, at hand) scaled by the mean inertia
      // stands in for it — both only detect an already converged iterate (tolerance 1e-8)
      float improvement = (prev_cost - cost) / scale;
      float gradient = sqrtf(m.meaninertia * fmaxf(gn, 0.f)) / scale;
      if (improvement < m.tolerance || gradient < m.tolerance) break;
    }
    float cost_new = tmw_linesearch(c, K, gauss);
    TMW_TICK(7);
    TMW_COUNT(36, 1000);
    TMW_FOR { const float g0 = L[K.l_Mgrad + lane], g1 = L[K.l_Mgrad + TMW_I1(K)]; c.wp0[TMW_LI] = g0; if (TMW_OK1(K)) c.wp1[TMW_LI] = g1; }      // (lane-local: the lane that wrote w_i reads it back)
    prev_cost = cost;
    cost = cost_new;
    float den = gn;
    // `last_if`: the loop is about to end — the iteration cap, or the improvement test of its next pass, which needs nothing but the two
    // costs at hand (in float32 a converged solve ends by that test: the gradient test sits below the rounding noise of gn).  The new
    // gradient's D^-1 N^T half, its norms and the next search direction would never be read; only qfrc_constraint is (Euler's right-hand
    // side).  tmw_update_gradient then returns -1 after J^T f (decided there, so that no flag lives across that product).
#ifndef TMW_CG_NO_EARLY_EXIT
    const float last_if = it + 1 >= m.iterations ? INFINITY : m.tolerance;     // ends if  (prev_cost - cost) / scale < last_if: the loop's own test
#else
    const float last_if = -INFINITY;
#endif
    gn = tmw_update_gradient(c, K, num, (prev_cost - cost) / scale, last_if);
    if (gn >= 0.f) {
      float beta = fmaxf(0.f, num / fmaxf(TM_MINVAL, den));
      TMW_FOR {
        const int i1 = TMW_I1(K);
        const float g0 = L[K.l_Mgrad + lane], g1 = L[K.l_Mgrad + i1], s0 = L[K.l_mv + lane], s1 = L[K.l_mv + i1];
        L[K.l_mv + lane] = -g0 + beta * s0; if (TMW_OK1(K)) L[K.l_mv + i1] = -g1 + beta * s1;
      }
      TMW_SYNC();
    }
    TMW_TICK(8);
  }
  TMW_STATS(K) += 64.f * (float)it;
  TMW_FOR { for (int i = lane; i < K.nv; i += 64) WST(m.s_warm, i) = TMW_QA(i); }
  if (c.dump) { TMW_FOR { for (int i = lane; i < K.nv; i += 64) c.dump[(size_t)(m.w_qacc + i) * (size_t)c.n + (size_t)c.e] = TMW_QA(i); } }
  // (tests: qfrc_constraint shares its words with l_search = l_hdamp in the lean layout — Euler overwrites them)
  if (c.dump) { TMW_FOR { for (int i = lane; i < K.nv; i += 64) c.dump[(size_t)(m.w_qfrc_constraint + i) * (size_t)c.n + (size_t)c.e] = L[K.l_qfrc_constraint + i]; } }
  TMW_SYNC();
}

// ------------------------------------------------------------------------------------------ forward / euler
TM_DEV void tmw_forward(WCtx &c, const WLayout &K, bool emit) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  TMW_TICK(12);
#ifndef TMW_STOP
#define TMW_STOP 99      // instruction-count experiments (tools/scratch/valu_phases.sh): leave the substep after phase TMW_STOP
#endif
  tmw_position(c, K, emit);
  TMW_TICK(0);
  if (TMW_STOP <= 0) return;
  tmw_velocity_inertia(c, K);
  TMW_TICK(1);
  if (TMW_STOP <= 1) return;
  if (K.m_spilled()) {
    // M is about to be factorised in place: keep a copy in global memory for Euler's factorisation of M + h D, and take the one
    // product with M the solver needs (M * warm start, tmw_solve_cg) now
    // (the warm start comes from the env's global record into the solver's iterate vector — region A is free here, l_Ma below is written
    // by the product anyway; its load is in flight across the copy loop.  Keeping it in two registers across the substeps, like qfrc_smooth,
    // was tried: 172 VGPRs, i.e. two instead of three waves per SIMD)
    TmwModel &m = *c.mp;
    TMW_REG(float, w0); TMW_REG(float, w1);
    TMW_FOR { w0[TMW_LI] = WST(m.s_warm, lane); w1[TMW_LI] = lane + 64 < K.nv ? WST(m.s_warm, lane + 64) : 0.f; }
    TMW_FOR {      // (blocks of six LDS loads in flight: one load / wait / store per trip was 18 serial LDS round trips)
      constexpr int NB = 6;
      for (int i0 = lane; i0 < K.nnz; i0 += 64 * NB) {
        float v[NB];
#pragma unroll
        for (int u = 0; u < NB; u++) { const int i = i0 + 64 * u; v[u] = L[K.l_M + (i < K.nnz ? i : K.nnz - 1)]; }
#pragma unroll
        for (int u = 0; u < NB; u++) { const int i = i0 + 64 * u; if (i < K.nnz) c.mspill[i] = v[u]; }
      }
    }
    TMW_FOR { L[K.l_qacc + lane] = w0[TMW_LI]; if (lane + 64 < K.nv) L[K.l_qacc + lane + 64] = w1[TMW_LI]; }
    TMW_SYNC();
    tmw_mul_m(c, K, K.l_qacc, K.l_Ma);
  }
  if (K.chains) tmw_factor_chains<false>(c, K, 0.f, -1); else tmw_factor(c, K, 0.f);
  TMW_TICK(2);
  if (TMW_STOP <= 2) return;
  if (K.chains) tmw_invert_chains(c, K); else tmw_invert_l(c, K);
  TMW_TICK(3);
  if (TMW_STOP <= 3) return;
  tmw_make_constraint(c, K);
  TMW_TICK(4);
  if (TMW_STOP <= 4) return;
  TMW_FOR { L[K.l_qacc_smooth + lane] = TMW_QFS(lane); if (TMW_OK1(K)) L[K.l_qacc_smooth + lane + 64] = TMW_QFS(lane + 64); }
  TMW_SYNC();
  tmw_solve(c, K, K.l_qacc_smooth, K.l_Mgrad);
  // (tests: qacc_smooth shares its words with the CG's search vector in the lean layout — copied out while it is there)
  if (c.dump) { TMW_FOR { for (int i = lane; i < K.nv; i += 64) c.dump[(size_t)(m.w_qacc_smooth + i) * (size_t)c.n + (size_t)c.e] = L[K.l_qacc_smooth + i]; } }
  TMW_TICK(5);
  tmw_solve_cg(c, K);
}
TM_DEV float tmw_euler(WCtx &c, const WLayout &K, float time) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
  float h = m.timestep;
  // chain layout: the matrix region holds L^-1 of the solver stage, which is dead now; M comes back from the env's global copy in one sweep
  // (all loads of a lane in flight together: one memory latency for the whole matrix).  The loads are issued FIRST, so that the right-hand side,
  // timestep * damping and the activation update below run under their latency; Euler then factorises M + h D in place again
  constexpr int MAXU = 20;                       // nnz <= 1280 (model_host.h)
  float mr[TMW_NL][MAXU];
  if (K.m_spilled()) {
    TMW_FOR {
#pragma unroll
      for (int u = 0; u < MAXU; u++) { int i = lane + 64 * u; if (64 * u < K.nnz) mr[TMW_LI][u] = c.mspill[i < K.nnz ? i : K.nnz - 1]; }
    }
  }
  // the right-hand side goes to l_Mgrad, dead since the CG loop ended; it is read here, lane by lane, BEFORE timestep * damping below overwrites
  // the words l_qfrc_constraint shares with l_search = l_hdamp in the lean layout
  TMW_FOR { const int i1 = TMW_I1(K); const float f0 = L[K.l_qfrc_constraint + lane], f1 = L[K.l_qfrc_constraint + i1]; L[K.l_Mgrad + lane] = TMW_QFS(lane) + f0; if (TMW_OK1(K)) L[K.l_Mgrad + i1] = TMW_QFS(lane + 64) + f1; }
  // timestep * damping (the diagonal Euler adds to M) into the dead search vector; the activation state is advanced here already — nothing
  // reads act between tmw_velocity_inertia and the end of the substep — so that ctrl's global load sits next to the loads above
  TMW_FOR {
    { const int i1 = TMW_I1(K); const float dm0 = m.dof_damping[lane], dm1 = m.dof_damping[i1]; L[K.l_hdamp + lane] = m.timestep * dm0; if (TMW_OK1(K)) L[K.l_hdamp + i1] = m.timestep * dm1; }
    for (int a = lane; a < K.nu; a += 64) {
      // (record mode: the action rows were transposed in behind the state + output rows)
      float ctrl = c.rs ? c.st[(size_t)c.e * (size_t)c.rs + (size_t)(m.s_prev_ctrl + a)] : (c.action ? c.action[(size_t)a * c.n + c.e] : 0.f);
      ctrl = fminf(fmaxf(ctrl, m.act_ctrlrange[a][0]), m.act_ctrlrange[a][1]);
      const float act = TMW_ACT(a);
      TMW_ACT(a) = act + ((ctrl - act) / fmaxf(TM_MINVAL, m.act_tau[a])) * h;
    }
  }
  if (K.m_spilled()) {
    TMW_FOR {
#pragma unroll
      for (int u = 0; u < MAXU; u++) { int i = lane + 64 * u; if (64 * u < K.nnz && i < K.nnz) L[K.l_M + i] = mr[TMW_LI][u]; }
    }
  }
  TMW_SYNC();
  TMW_TICK(8);
  if (K.chains) tmw_factor_chains<true>(c, K, h, K.l_Mgrad); else tmw_factor(c, K, h, K.l_Mgrad);
  TMW_TICK(9);
  if (K.chains) tmw_subst_chains(c, K, K.l_Mgrad); else tmw_subst_down(c, K, K.l_Mgrad);
  TMW_TICK(11);
  TMW_FOR {
    { const int i1 = TMW_I1(K); const float v0 = L[K.l_qvel + lane], v1 = L[K.l_qvel + i1], g0 = L[K.l_Mgrad + lane], g1 = L[K.l_Mgrad + i1]; L[K.l_qvel + lane] = v0 + g0 * h; if (TMW_OK1(K)) L[K.l_qvel + i1] = v1 + g1 * h; }
  }
  TMW_SYNC();
  TMW_FOR {
    for (int j = lane; j < K.njnt; j += 64) {
      int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      if (m.jnt_type[j] == 0) {
        for (int k = 0; k < 3; k++) L[K.l_qpos + qa + k] += h * L[K.l_qvel + da + k];
        float v[3] = {L[K.l_qvel + da + 3], L[K.l_qvel + da + 4], L[K.l_qvel + da + 5]}, q[4], qr[4], q2[4];
        for (int k = 0; k < 4; k++) q[k] = L[K.l_qpos + qa + 3 + k];
        float nn = tm_normalize3(v), ang = h * nn * 0.5f, sn, cs;
        sincosf(ang, &sn, &cs);
        qr[0] = cs; qr[1] = v[0] * sn; qr[2] = v[1] * sn; qr[3] = v[2] * sn;
        tm_quat_mul(q2, q, qr);
        tm_normalize4(q2);
        for (int k = 0; k < 4; k++) L[K.l_qpos + qa + 3 + k] = q2[k];
      } else {
        L[K.l_qpos + qa] += h * L[K.l_qvel + da];
      }
    }
  }
  TMW_SYNC();
  return time + h;
}

// ------------------------------------------------------------------------------------------ debug dump (tests)
// copies solver-stage intermediates of the last forward from LDS to the lane-per-env workspace rows so that the same
// named arrays (tmjx_debug_rows) can be compared with the oracle; never called on the timed path
TM_DEV void tmw_dump(WCtx &c, const WLayout &K, float *ws) {
  TmwModel &m = *c.mp; float *L = c.L; TMW_LANE_DECL
#define WDUMP(row, i) ws[(size_t)((row) + (i)) * (size_t)c.n + (size_t)c.e]
  TMW_FOR {
    for (int i = lane; i < K.nnz; i += 64) WDUMP(m.w_M, i) = K.m_spilled() ? c.mspill[i] : L[K.l_M + i];
    for (int i = lane; i < K.nv * 6; i += 64) WDUMP(m.w_cdof, i) = L[K.l_cdof + i];
    for (int i = lane; i < K.nv; i += 64) {
      WDUMP(m.w_qfrc_smooth, i) = TMW_QFS(i);        // (qacc_smooth, qfrc_constraint: written where they arise — tmw_forward, tmw_solve_cg)
    }
    for (int i = lane; i < K.ncon; i += 64) WDUMP(m.w_con_dist, i) = L[K.l_con_dist + i];
    for (int cc = lane; cc < K.ncon; cc += 64) {
      float fr[9];
      tmw_get_con_frame(L, K, cc, fr);
      tm_cross(fr + 6, fr, fr + 3);
      for (int k = 0; k < 9; k++) WDUMP(m.w_con_frame, cc * 9 + k) = fr[k];
    }
    for (int r = lane; r < K.nefc; r += 64) {   // efc_D / efc_aref were written by tmw_make_constraint; force per ORIGINAL row
      int kr = -1;
      if (r < K.nlim) { float sv = (float)TMW_LIMSIGN(K)[r]; if (sv != 0.f) kr = (int)fabsf(sv) - 1; }
      else if (TMW_CCROW(K)[(r - K.nlim) >> 2] != 255) kr = TMW_CCROW(K)[(r - K.nlim) >> 2] + ((r - K.nlim) & 3);
      float ja = kr >= 0 ? L[K.l_Jaref + kr] : 0.f;
      WDUMP(m.w_efc_force, r) = ja < 0.f ? -L[K.l_efc_D + TMW_DIDX(kr)] * ja : 0.f;
    }
    if (lane < 3) WDUMP(m.w_com, lane) = L[K.l_com + lane];
    // solver statistics of the last substep: CG iterations, line-search iterations (summed), rows that entered the solver, of which limits
    if (lane < 4) { int pk = (int)TMW_STATS(K); WDUMP(m.w_solver_stats, lane) = (float)(lane == 0 ? pk >> 6 : lane == 1 ? (pk & 63) : lane == 2 ? c.nact : c.nla); }
    // which ORIGINAL constraint rows entered the solver (violated limits, the four pyramid rows of penetrating contacts): 1 / 0
    for (int r = lane; r < K.nefc; r += 64) {
      bool in = r < K.nlim ? TMW_LIMSIGN(K)[r] != 0 : TMW_CCROW(K)[(r - K.nlim) >> 2] != 255;
      WDUMP(m.w_efc_in, r) = in ? 1.f : 0.f;
    }
  }
#undef WDUMP
  TMW_SYNC();
}
