// csrc/wave_layout.h — dimensions and LDS map of the wave-per-env physics kernel as ONE constexpr-constructible value.
//
// The kernel is instantiated twice: with `constexpr WLayout k(rodent dims)` every LDS offset and loop bound folds into
// an instruction immediate (no scalar loads of layout fields inside the pivot / mat-vec loops — each such s_load also
// forces an lgkmcnt(0) wait that serialises the LDS pipeline), and with a run-time WLayout built from the model's
// dims for any other model.  The host picks the specialised kernel when the dims match (tmjx_hip.hip).
#pragma once

struct WLayout {
  int nbody, njnt, nq, nv, nu, ncon, nlim, nefc, ngroup, nnz, nphys, nround_body, nround_dof;
  // persistent part
  int l_qpos, l_qvel, l_act, l_cdof, l_M, l_con_dist, l_con_off, l_con_frame, l_lim_sign, l_qfrc_smooth,
      l_com, l_sv, l_wr, l_tdof, l_tgrp, l_hdamp, l_con_mu, l_con_grpb, l_rowmap, l_ccrow, l_dummy, l_alias0;
  // aliased region A (solver stage)
  int l_LD, l_Dinv, l_efc_D, l_efc_aref, l_Jaref, l_jv, l_qacc_smooth, l_qacc, l_Ma, l_Mgrad, l_search, l_mv,
      l_qfrc_constraint;
  // aliased region B (position / velocity stage), same base as region A
  int l_scanA, l_scanB, l_jl_anchor, l_jl_axis, l_xipos, l_cinert, l_cfrc, l_dscanA, l_dscanB;
  int lds_floats;
  int lean;     // 1: chain layout with the matrix spilled (m_spilled()): act, qfrc_smooth in global memory, shared contact normal
  int chains;   // 0: any tree (LDS-resident sparse factorisation); 1: the rodent's dof chains (register-resident, wave_physics.h)

  constexpr WLayout(int nb, int nj, int nq_, int nv_, int nu_, int nc, int nl, int nnz_, int ng, int rb, int rd, int ch = 0)
      : nbody(nb), njnt(nj), nq(nq_), nv(nv_), nu(nu_), ncon(nc), nlim(nl), nefc(nl + 4 * nc), ngroup(ng), nnz(nnz_),
        nphys(nq_ + nv_ + nu_ + nv_ + 1), nround_body(rb), nround_dof(rd),
        l_qpos(0), l_qvel(0), l_act(0), l_cdof(0), l_M(0), l_con_dist(0), l_con_off(0), l_con_frame(0),
        l_lim_sign(0), l_qfrc_smooth(0), l_com(0), l_sv(0), l_wr(0), l_tdof(0), l_tgrp(0), l_hdamp(0), l_con_mu(0), l_con_grpb(0), l_rowmap(0), l_ccrow(0), l_dummy(0),
        l_alias0(0), l_LD(0), l_Dinv(0), l_efc_D(0), l_efc_aref(0), l_Jaref(0), l_jv(0), l_qacc_smooth(0), l_qacc(0), l_Ma(0),
        l_Mgrad(0), l_search(0), l_mv(0), l_qfrc_constraint(0), l_scanA(0), l_scanB(0), l_jl_anchor(0),
        l_jl_axis(0), l_xipos(0), l_cinert(0), l_cfrc(0), l_dscanA(0), l_dscanB(0), lds_floats(0), lean(ch && 2 * nb * 8 <= nnz_ && nv_ * 7 <= nb * 8 && nu_ <= ng * 6 && nv_ * 7 + nb * 6 <= nnz_), chains(ch) {
    int l = 0;
    // NOT here (they were, 295 words): the warm start, ctrl, act_dot, qfrc_actuator and timestep * damping.  LDS is granted in 1280-byte
    // granules on gfx950 (tools/micro/lds_occupancy.hip): 16 284 bytes took 13 of the 128 granules of a CU, i.e. NINE resident envs, not the
    // ten that "16 KB x 10 = 160 KB" suggests; 15 104 bytes take 12 -> ten.  Those five are touched once or twice per substep: they live in
    // the env's global record (warm start, qfrc_actuator), are re-read from it (ctrl), recomputed (act_dot) or staged in a solver vector
    // that is dead by then (l_hdamp = l_search during Euler).  The CHAIN layout (the rodent) goes one granule further, 14 040 bytes = 11 of
    // 128 -> ELEVEN envs: the activation state and qfrc_smooth live in global memory too (the env's record / the tail of its inertia-matrix
    // scratch; two and five accesses per substep), the contact frames share their first row (every contact is against the one static floor
    // plane: model_host.h checks it), the violated-limit table is bytes, the paw-group table one word per group.
    l_qpos = l; l += nq; l_qvel = l; l += nv; l_act = l; l += lean ? 0 : nu;
    l_cdof = l; l += nv * 6; l_M = l; l += nnz; l_con_dist = l; l += ncon; l_con_off = l; l += ncon * 3;
    l_con_frame = l; l += lean ? 3 + ncon * 3 : ncon * 6; l_lim_sign = l; l += (nlim + 3) / 4; l_qfrc_smooth = l; l += lean ? 0 : nv;
    l_com = l; l += 4; l_sv = l; l += ngroup * 6;
    // per-dof index table: two words per dof — or (lean) NOT in LDS: one packed word per dof in two registers of the lane that owns the dof (round 5;
    // every read of the table in the chain kernels is lane-local: DModel::tpack, WCtx::tp0 / tp1)
    l_tdof = l; l += lean ? 0 : nv * 2;
    l_tgrp = l; l += lean ? ngroup : (ngroup + 3) / 4;      // bytes: last dof of each paw group (-1 = none); lean: one packed word per group (DModel::gpack)
    l_con_mu = l; l += lean ? 0 : ncon;  // friction coefficient of every contact slot: a model constant the products with J / J^T need on every call (lean: ONE
                                         // coefficient for all slots, a uniform read of the model — model_host.h checks it)
    // byte tables: compact row -> original row of the ACTIVE constraint rows (wave_physics.h: tmw_make_constraint); contact -> its first
    // compact row (255 = none) and, behind it in the same array, contact -> paw group (a model constant)
    l_rowmap = l; l += (nefc + 3) / 4; l_ccrow = l; l_con_grpb = l; l += (2 * nc + 3) / 4;
    l = (l + 3) & ~3;
    l_alias0 = l;
    l_LD = l; l += nnz; l_Dinv = l; l += nv; l_efc_D = l; l += nefc; l_efc_aref = l; l += nefc; l_Jaref = l; l += nefc;
    // liveness-based sharing: aref is dead once the CG start point is chosen -> J*search (jv) re-uses it; the per-contact
    // wrenches of J^T f (wr, 6*ncon <= nefc) are only live inside tmw_jt_force, never together with jv
    l_jv = l_efc_aref; l_wr = l_efc_aref;
    l_qacc_smooth = l; l += nv; l_qacc = l; l += nv; l_Ma = l; l += nv;
    l_Mgrad = l; l += nv; l_search = l; l += nv; l_hdamp = l_search; l_mv = l; l += nv; l_qfrc_constraint = l; l += nv;
    l_dummy = l_mv;   // per-lane source of masked LDS READS (never written through; needs 64 <= nv words)
    int endA = l;
    l = l_alias0;
    l_scanA = l; l += nbody * 8; l_scanB = l; l += nbody * 8; l_jl_anchor = l; l += njnt * 3; l_jl_axis = l; l += njnt * 3;
    l_xipos = l; l += nbody * 3;
    int endB1 = l;
    // velocity stage: behind scanA (which still holds the world transforms); scanB / joint frames / xipos are dead by then
    l = l_alias0 + nbody * 8;
    l_cinert = l; l += nbody * 10; l_cfrc = l; l += nbody * 6; l_dscanB = l; l += nv * 7;
    l_dscanA = l_alias0;   // dof-scan buffer A overlays the (by then dead) body transforms: needs nv*7 <= nbody*8
    int endB2 = l;
    lds_floats = endA > endB1 ? endA : endB1;
    if (endB2 > lds_floats) lds_floats = endB2;
    if (ch && 2 * nbody * 8 <= nnz && nv * 7 <= nbody * 8) {
      // Chain layout (register-resident factorisation, wave_physics.h): ONE matrix region.  M is factorised IN PLACE (a chain's
      // rows are in registers before its first row of L is stored), so l_LD == l_M; M itself goes to a per-env global scratch right
      // after it is built (Euler's second factorisation reads it back from there, M * warm start is taken before the first one).
      // Between Euler and the next "M rows" step the region is dead and hosts the transform / dof scan buffers.  With that the
      // solver-stage region A loses LD and the kinematics region B its scan buffers: 15.9 KB per env instead of 20.4 KB (13 instead
      // of 17 LDS granules of 1280 bytes: 9 instead of 7 envs per CU; the cuts listed above bring it to 11 granules).
      l_LD = l_M;
      l_scanA = l_M; l_scanB = l_M + nbody * 8; l_dscanA = l_scanA; l_dscanB = l_scanB;
      l = l_alias0;
#ifdef TMW_NO_EFCD_PACK
      l_Dinv = l; l += nv; l_efc_D = l; l += nefc;
#else
      l_Dinv = l; l += nv; l_efc_D = l; l += lean ? nlim + ncon : nefc;
#endif
      l_efc_aref = l; l += nefc; l_Jaref = l; l += nefc;      // (lean: efc_D per CONTACT, wave_physics.h TMW_DIDX)
      l_jv = l_efc_aref; l_wr = l_efc_aref;
      // Solver vectors (round 4: 3 520 -> 3 199 words = 12 796 bytes = TEN granules of 1 280 bytes -> TWELVE envs per CU).  What left LDS or
      // shares words: D = 1 / Dinv and the previous CG direction's w live in two registers per lane (WCtx::dg*, wp*: read lane-locally only);
      // qacc_smooth (dead once the CG's start point is chosen) and qfrc_constraint (written by the LAST J^T f of a CG pass, after that pass's
      // line search has consumed search_q; read by D^-1 N^T and, after the loop, by Euler's right-hand side before timestep * damping is
      // staged there) share the words of l_search; Euler's right-hand side rides in l_Mgrad; the friction coefficient is a model scalar.
      // Round 5: the CG's iterate qacc and ut = y - y_s, which every pass of the loop updates lane-locally, live in two registers per lane each
      // (WCtx::qa*, ma*); the LDS image a product needs across lanes at the START of the solve (J qacc, M qacc, D^-1 N^T ut) is staged through
      // the search direction's and its gradient's words, dead until the first gradient: l_qacc = l_mv, l_Ma = l_Mgrad.
#ifdef TMW_NO_QA_REGS
      l_qacc = l; l += nv; l_Ma = l; l += nv;
      l_Mgrad = l; l += nv; l_search = l; l += nv; l_hdamp = l_search; l_mv = l; l += nv;
#else
      if (!lean) { l_qacc = l; l += nv; l_Ma = l; l += nv; }
      l_Mgrad = l; l += nv; l_search = l; l += nv; l_hdamp = l_search; l_mv = l; l += nv;
      if (lean) { l_qacc = l_mv; l_Ma = l_Mgrad; }
#endif
      l_qacc_smooth = lean ? l_search : l; l += lean ? 0 : nv; l_qfrc_constraint = lean ? l_search : l; l += lean ? 0 : nv;
      l_dummy = l_mv;
      int eA = l;
      // region B: xipos | joint anchors | joint axes; cinert overlays the joint frames (dead once cdof is built) but not xipos,
      // which the cinert step still reads.  cfrc (lean) lives in the MATRIX region behind the dof-scan buffer (dead between Euler and the
      // "M rows" step, which takes its bias forces into registers before it writes the first row of M); else behind cinert
      l = l_alias0;
      l_xipos = l; l += nbody * 3; l_jl_anchor = l; l += njnt * 3; l_jl_axis = l; l += njnt * 3;
      int eB1 = l;
      // (lean, round 5: cinert overlays xipos too — the cinert step takes its bodies' origins into registers before the first row is written)
#ifdef TMW_NO_XIPOS_OVERLAY
      l_cinert = l_alias0 + nbody * 3; l_cfrc = lean ? l_M + nv * 7 : l_cinert + nbody * 10;
#else
      l_cinert = lean ? l_alias0 : l_alias0 + nbody * 3; l_cfrc = lean ? l_M + nv * 7 : l_cinert + nbody * 10;
#endif
      int eB2 = lean ? l_cinert + nbody * 10 : l_cfrc + nbody * 6;
      lds_floats = eA > eB1 ? eA : eB1;
      if (eB2 > lds_floats) lds_floats = eB2;
    }
  }
  // true when M lives in the per-env global scratch outside the "M rows" -> first factorisation window (see above)
  constexpr bool m_spilled() const { return chains && l_LD == l_M; }
};

// the rodent walker of the reference (track_mjx/environment/walker/assets/rodent/rodent.xml): 68 bodies, 68 joints,
// nq 74, nv 73, 38 actuators, 30 contact slots, 67 joint limits, 1119 non-zeros in the tree-sparse M, 8 paw bodies
#define TMW_RODENT_DIMS 68, 68, 74, 73, 38, 30, 67, 1119, 8, 6, 6
// its dof tree: a trunk chain (dofs 0..11: the free joint, then the 6 dofs up to the pelvis/lumbar branch point) and six leaf
// chains X(first dof, length, depth of the first dof) of consecutive dofs; listed in the order they are eliminated
#define TMW_RODENT_TRUNK 12
#define TMW_RODENT_LEAF_CHAINS(X) X(65, 8, 6) X(57, 8, 6) X(48, 9, 6) X(24, 24, 12) X(18, 6, 12) X(12, 6, 12)
#define TMW_RODENT_NCHAIN 6
// depth of a dof and address of its row in the tree-sparse storage (rows back to back in dof order, row i = depth+1 words),
// folded at compile time from the chain table; model_host.h checks the loaded model against it
constexpr int tmw_chain_depth(int i) {
  if (i < TMW_RODENT_TRUNK) return i;
#define TMW_X(first, n, d0) if (i >= first && i < first + n) return d0 + i - first;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  return 0;
}
constexpr int tmw_chain_madr(int i) { int a = 0; for (int j = 0; j < i; j++) a += tmw_chain_depth(j) + 1; return a; }
constexpr int tmw_chain_maxdepth(int i0, int i1) { int d = 0; for (int i = i0; i < i1; i++) if (tmw_chain_depth(i) > d) d = tmw_chain_depth(i); return d; }
constexpr int tmw_chain_first(int i) {
  if (i < TMW_RODENT_TRUNK) return 0;
#define TMW_X(first, n, d0) if (i >= first && i < first + n) return first;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  return 0;
}
// longest run of in-chain ancestors (i - chain start) among dofs i0 .. i1-1
constexpr int tmw_chain_maxrun(int i0, int i1) { int r = 0; for (int i = i0; i < i1; i++) if (i - tmw_chain_first(i) > r) r = i - tmw_chain_first(i); return r; }
// number of leaf-chain rows eliminated BEFORE the chain starting at dof `first` (chains in list order): position of the chain's
// rows in the 4-deep operand queue of the MFMA Schur accumulation (wave_physics.h)
constexpr int tmw_chain_rows_before(int first) {
  int n = 0;
#define TMW_X(f, len, d0) if (f == first) return n; n += len;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  return n;
}
