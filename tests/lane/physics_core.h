// tests/lane/physics_core.h — TEST INFRASTRUCTURE (lives under tests/, never compiled into the product library): lane-per-env formulation of K2 (one full `mjx.step` per lane, everything through global memory): the first
// HIP implementation, kept as an independent cross-check of the wave-per-env product kernel (wave_physics.h).  NOT part of the
// product library: compiled only with -DTMJX_LANE_IMPL (tests/lane/libtmjx_hip_lane.so) and by the host emulation.
//
// Replaces (reference call site) track_mjx/environment/task/single_clip_tracking.py:219
//   pipeline_step -> brax.mjx.pipeline.step -> mujoco.mjx.step   (mujoco-mjx 3.3.2, third-party).
// Same algorithm as MJX (kinematics, com_pos, crb, factor, collision, make_constraint, com_vel, passive,
// rne, actuation, acceleration, CG solve with Newton bracketing line search, Euler with implicit joint
// damping) but re-formulated for the GPU instead of translated:
//   * tree-sparse inertia matrix (1119 non-zeros instead of dense 73x73) with in-place L^T D L,
//   * matrix-free constraint Jacobian: limit rows are one-hot; the 4 pyramid rows of a contact are
//     evaluated from the paw body's spatial velocity / accumulated as a body wrench (no 187x73 efc_J),
//   * every per-env array is a row of an env-minor buffer: lane == env, all accesses coalesced.
// This file holds the device functions only; launch code is in tmjx_hip.hip.
#pragma once
#include "../../track_mjx_amd/csrc/tm_common.h"

// ------------------------------------------------------------------------------------------ fwd_position
// smooth.kinematics + smooth.com_pos (subtree COM of the single moving tree, cinert, cdof)
TM_DEV void tm_kinematics_com(const DModel &m, EnvRef r) {
  for (int k = 0; k < 3; k++) ST(m.s_xpos, k) = 0.f;
  WS(m.w_xquat, 0) = 1.f; WS(m.w_xquat, 1) = 0.f; WS(m.w_xquat, 2) = 0.f; WS(m.w_xquat, 3) = 0.f;
  for (int k = 0; k < 3; k++) WS(m.w_xipos, k) = 0.f;
  float csum[3] = {0.f, 0.f, 0.f}, msum = 0.f;
  for (int b = 1; b < m.nbody; b++) {
    int p = m.body_parentid[b];
    float ppos[3], pq[4], pos[3], quat[4], t[3];
    TM_LD(ppos, ST, m.s_xpos, p * 3, 3);
    TM_LD(pq, WS, m.w_xquat, p * 4, 4);
    tm_rotate(t, m.body_pos[b], pq);
    for (int k = 0; k < 3; k++) pos[k] = ppos[k] + t[k];
    tm_quat_mul(quat, pq, m.body_quat[b]);
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
      int j = m.body_jntadr[b] + jj, qa = m.jnt_qposadr[j];
      if (m.jnt_type[j] == 0) {
        TM_LD(pos, ST, m.s_qpos, qa, 3);
        TM_LD(quat, ST, m.s_qpos, qa + 3, 4);
        tm_normalize4(quat);
        TM_SV(ST, m.s_qpos, qa + 3, quat, 4);  // kinematics also normalises the stored quaternion
        TM_SV(WS, m.w_xanchor, j * 3, pos, 3);
        WS(m.w_xaxis, j * 3) = 0.f; WS(m.w_xaxis, j * 3 + 1) = 0.f; WS(m.w_xaxis, j * 3 + 2) = 1.f;
      } else {
        float anchor[3], axis[3], ql[4], q2[4];
        tm_rotate(t, m.jnt_pos[j], quat);
        for (int k = 0; k < 3; k++) anchor[k] = t[k] + pos[k];
        tm_rotate(axis, m.jnt_axis[j], quat);
        float ang = (ST(m.s_qpos, qa) - m.qpos0[qa]) * 0.5f, sn = sinf(ang), cs = cosf(ang);
        ql[0] = cs; ql[1] = m.jnt_axis[j][0] * sn; ql[2] = m.jnt_axis[j][1] * sn; ql[3] = m.jnt_axis[j][2] * sn;
        tm_quat_mul(q2, quat, ql);
        for (int k = 0; k < 4; k++) quat[k] = q2[k];
        tm_rotate(t, m.jnt_pos[j], quat);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - t[k];
        TM_SV(WS, m.w_xanchor, j * 3, anchor, 3);
        TM_SV(WS, m.w_xaxis, j * 3, axis, 3);
      }
    }
    TM_SV(ST, m.s_xpos, b * 3, pos, 3);
    TM_SV(WS, m.w_xquat, b * 4, quat, 4);
    float ip[3];
    tm_rotate(t, m.body_ipos[b], quat);
    for (int k = 0; k < 3; k++) ip[k] = pos[k] + t[k];
    TM_SV(WS, m.w_xipos, b * 3, ip, 3);
    if (m.body_moving[b]) { float mb = m.body_mass[b]; for (int k = 0; k < 3; k++) csum[k] += ip[k] * mb; msum += mb; }
    if (b == m.torso_idx) { float X[9]; tm_quat_to_mat(X, quat); TM_SV(ST, m.s_xmat_torso, 0, X, 9); }
  }
  float com[3];
  for (int k = 0; k < 3; k++) com[k] = csum[k] / msum;
  TM_SV(WS, m.w_com, 0, com, 3);
  // cinert (frame centred at the tree's COM) and crb initialisation
  for (int b = 0; b < m.nbody; b++) {
    float c[10];
    if (!m.body_moving[b]) { for (int k = 0; k < 10; k++) c[k] = 0.f; }
    else {
      float q[4], bq[4], X[9], ip[3], off[3];
      TM_LD(bq, WS, m.w_xquat, b * 4, 4);
      tm_quat_mul(q, bq, m.body_iquat[b]);
      tm_quat_to_mat(X, q);
      TM_LD(ip, WS, m.w_xipos, b * 3, 3);
      float mass = m.body_mass[b];
      for (int k = 0; k < 3; k++) off[k] = ip[k] - com[k];
      const float *in = m.body_inertia[b];
      float oo = tm_dot3(off, off);
      float I00 = X[0] * in[0] * X[0] + X[1] * in[1] * X[1] + X[2] * in[2] * X[2];
      float I11 = X[3] * in[0] * X[3] + X[4] * in[1] * X[4] + X[5] * in[2] * X[5];
      float I22 = X[6] * in[0] * X[6] + X[7] * in[1] * X[7] + X[8] * in[2] * X[8];
      float I01 = X[0] * in[0] * X[3] + X[1] * in[1] * X[4] + X[2] * in[2] * X[5];
      float I02 = X[0] * in[0] * X[6] + X[1] * in[1] * X[7] + X[2] * in[2] * X[8];
      float I12 = X[3] * in[0] * X[6] + X[4] * in[1] * X[7] + X[5] * in[2] * X[8];
      c[0] = I00 + (oo - off[0] * off[0]) * mass; c[1] = I11 + (oo - off[1] * off[1]) * mass;
      c[2] = I22 + (oo - off[2] * off[2]) * mass;
      c[3] = I01 - off[0] * off[1] * mass; c[4] = I02 - off[0] * off[2] * mass; c[5] = I12 - off[1] * off[2] * mass;
      c[6] = off[0] * mass; c[7] = off[1] * mass; c[8] = off[2] * mass; c[9] = mass;
    }
    TM_SV(WS, m.w_cinert, b * 10, c, 10);
    TM_SV(WS, m.w_crb, b * 10, c, 10);
  }
  // cdof: motion axes in the global frame centred at the COM
  for (int j = 0; j < m.njnt; j++) {
    int b = m.jnt_bodyid[j], da = m.jnt_dofadr[j];
    float anchor[3], off[3];
    TM_LD(anchor, WS, m.w_xanchor, j * 3, 3);
    for (int k = 0; k < 3; k++) off[k] = com[k] - anchor[k];
    if (m.jnt_type[j] == 0) {
      float bq[4], X[9];
      TM_LD(bq, WS, m.w_xquat, b * 4, 4);
      tm_quat_to_mat(X, bq);
      for (int rr = 0; rr < 3; rr++) {
        for (int k = 0; k < 6; k++) WS(m.w_cdof, (da + rr) * 6 + k) = (k == 3 + rr) ? 1.f : 0.f;
        float ax[3] = {X[rr], X[3 + rr], X[6 + rr]}, c[3];
        tm_cross(c, ax, off);
        TM_SV(WS, m.w_cdof, (da + 3 + rr) * 6, ax, 3);
        TM_SV(WS, m.w_cdof, (da + 3 + rr) * 6 + 3, c, 3);
      }
    } else {
      float ax[3], c[3];
      TM_LD(ax, WS, m.w_xaxis, j * 3, 3);
      tm_cross(c, ax, off);
      TM_SV(WS, m.w_cdof, da * 6, ax, 3);
      TM_SV(WS, m.w_cdof, da * 6 + 3, c, 3);
    }
  }
}

// smooth.crb: composite inertias up the tree, then the tree-sparse M (row i = dof i, its ancestors)
TM_DEV void tm_crb(const DModel &m, EnvRef r) {
  for (int b = m.nbody - 1; b >= 1; b--) {
    int p = m.body_parentid[b];
    if (p == 0) continue;
    for (int k = 0; k < 10; k++) WS(m.w_crb, p * 10 + k) += WS(m.w_crb, b * 10 + k);
  }
  for (int i = 0; i < m.nv; i++) {
    float I[10], cd[6], buf[6];
    TM_LD(I, WS, m.w_crb, m.dof_bodyid[i] * 10, 10);
    TM_LD(cd, WS, m.w_cdof, i * 6, 6);
    tm_inert_mul(buf, I, cd);
    int adr = m.dof_Madr[i], k = 0;
    for (int j = i; j >= 0; j = m.dof_parentid[j], k++) {
      float cj[6];
      TM_LD(cj, WS, m.w_cdof, j * 6, 6);
      float s = buf[0] * cj[0] + buf[1] * cj[1] + buf[2] * cj[2] + buf[3] * cj[3] + buf[4] * cj[4] + buf[5] * cj[5];
      if (k == 0) s += m.dof_armature[i];
      WS(m.w_M, adr + k) = s;
    }
  }
}

// smooth.factor_m, tree-sparse: LD = L^T D L of (M + hdamp * diag(dof_damping)); Dinv = 1/D
TM_DEV void tm_factor(const DModel &m, EnvRef r, float hdamp) {
  for (int i = 0; i < m.nnz; i++) WS(m.w_LD, i) = WS(m.w_M, i);
  if (hdamp != 0.f) for (int i = 0; i < m.nv; i++) WS(m.w_LD, m.dof_Madr[i]) += hdamp * m.dof_damping[i];
  for (int k = m.nv - 1; k >= 0; k--) {
    int ak = m.dof_Madr[k];
    float inv = 1.f / WS(m.w_LD, ak);
    WS(m.w_Dinv, k) = inv;
    int mm = 1;
    for (int i = m.dof_parentid[k]; i >= 0; i = m.dof_parentid[i], mm++) {
      float a = WS(m.w_LD, ak + mm) * inv;
      int ai = m.dof_Madr[i], di = m.dof_depth[i];
      for (int p = 0; p <= di; p++) WS(m.w_LD, ai + p) -= a * WS(m.w_LD, ak + mm + p);
      WS(m.w_LD, ak + mm) = a;
    }
  }
}
// smooth.solve_m in place on workspace vector `x`
TM_DEV void tm_solve(const DModel &m, EnvRef r, int x) {
  for (int k = m.nv - 1; k >= 0; k--) {
    float xk = WS(x, k);
    int ak = m.dof_Madr[k], mm = 1;
    for (int i = m.dof_parentid[k]; i >= 0; i = m.dof_parentid[i], mm++) WS(x, i) -= WS(m.w_LD, ak + mm) * xk;
  }
  for (int k = 0; k < m.nv; k++) {
    float acc = WS(x, k) * WS(m.w_Dinv, k);
    int ak = m.dof_Madr[k], mm = 1;
    for (int i = m.dof_parentid[k]; i >= 0; i = m.dof_parentid[i], mm++) acc -= WS(m.w_LD, ak + mm) * WS(x, i);
    WS(x, k) = acc;
  }
}
// support.mul_m: y = M x (tree-sparse, symmetric)
TM_DEV void tm_mul_m(const DModel &m, EnvRef r, int x, int y) {
  for (int i = 0; i < m.nv; i++) WS(y, i) = 0.f;
  for (int i = 0; i < m.nv; i++) {
    int a = m.dof_Madr[i], k = 1;
    float xi = WS(x, i), acc = WS(m.w_M, a) * xi;
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j], k++) {
      float mij = WS(m.w_M, a + k);
      acc += mij * WS(x, j);
      WS(y, j) += mij * xi;
    }
    WS(y, i) += acc;
  }
}

// ------------------------------------------------------------------------------------------ collision
// collision_driver.collision for the static plane-vs-paw slots; stores dist, offset from the COM, frame
TM_DEV void tm_collision(const DModel &m, EnvRef r) {
  float com[3];
  TM_LD(com, WS, m.w_com, 0, 3);
  for (int c = 0; c < m.ncon; c++) {
    int b1 = m.con_body1[c], b2 = m.con_body2[c];
    float xp[3], xq[4], t[3], pp[3], pq[4], pm[9], gp[3], gq[4], gm[9];
    TM_LD(xp, ST, m.s_xpos, b1 * 3, 3); TM_LD(xq, WS, m.w_xquat, b1 * 4, 4);
    tm_rotate(t, m.con_g1_pos[c], xq);
    for (int k = 0; k < 3; k++) pp[k] = xp[k] + t[k];
    tm_quat_mul(pq, xq, m.con_g1_quat[c]); tm_quat_to_mat(pm, pq);
    TM_LD(xp, ST, m.s_xpos, b2 * 3, 3); TM_LD(xq, WS, m.w_xquat, b2 * 4, 4);
    tm_rotate(t, m.con_g2_pos[c], xq);
    for (int k = 0; k < 3; k++) gp[k] = xp[k] + t[k];
    tm_quat_mul(gq, xq, m.con_g2_quat[c]); tm_quat_to_mat(gm, gq);
    float n[3] = {pm[2], pm[5], pm[8]}, fr[9], pos[3], dist;
    const float *size = m.con_g2_size[c];
    if (m.con_type[c] == 3) {
      float axis[3] = {gm[2], gm[5], gm[8]}, b[3], na = tm_dot3(n, axis), cr[3];
      for (int k = 0; k < 3; k++) b[k] = axis[k] - n[k] * na;
      float bn = tm_normalize3(b);
      if (bn < 0.5f) { bool yy = (-0.5f < n[1]) && (n[1] < 0.5f); b[0] = 0.f; b[1] = yy ? 1.f : 0.f; b[2] = yy ? 0.f : 1.f; }
      tm_cross(cr, n, b);
      for (int k = 0; k < 3; k++) { fr[k] = n[k]; fr[3 + k] = b[k]; fr[6 + k] = cr[k]; }
      float sg = m.con_sub[c] == 0 ? 1.f : -1.f, end[3];
      for (int k = 0; k < 3; k++) end[k] = gp[k] + sg * (axis[k] * size[1]);
      tm_plane_sphere(n, pp, end, size[0], dist, pos);
    } else if (m.con_type[c] == 4) {
      float loc[3], sup[3], w[3], df[3];
      for (int k = 0; k < 3; k++) loc[k] = gm[k] * n[0] + gm[3 + k] * n[1] + gm[6 + k] * n[2];
      for (int k = 0; k < 3; k++) sup[k] = loc[k] * size[k];
      tm_normalize3(sup);
      for (int k = 0; k < 3; k++) sup[k] = -sup[k] * size[k];
      for (int k = 0; k < 3; k++) w[k] = gm[k * 3] * sup[0] + gm[k * 3 + 1] * sup[1] + gm[k * 3 + 2] * sup[2];
      for (int k = 0; k < 3; k++) { pos[k] = gp[k] + w[k]; df[k] = pos[k] - pp[k]; }
      dist = tm_dot3(n, df);
      for (int k = 0; k < 3; k++) pos[k] = pos[k] - n[k] * dist * 0.5f;
      tm_make_frame(n, fr);
    } else {
      tm_plane_sphere(n, pp, gp, size[0], dist, pos);
      tm_make_frame(n, fr);
    }
    WS(m.w_con_dist, c) = dist;
    for (int k = 0; k < 3; k++) WS(m.w_con_off, c * 3 + k) = pos[k] - com[k];
    TM_SV(WS, m.w_con_frame, c * 9, fr, 9);
  }
}

// ------------------------------------------------------------------------------------------ matrix-free J
// out[nefc] = J v.  Limit rows one-hot (sign 0/+1/-1 already active-masked); contact rows from the paw
// body's spatial velocity sv = sum_{dof in chain} cdof[dof] * v[dof].
TM_DEV void tm_jmul(const DModel &m, EnvRef r, int v, int out) {
  for (int l = 0; l < m.nlim; l++) WS(out, l) = WS(m.w_lim_sign, l) * WS(v, m.jnt_dofadr[m.lim_jnt[l]]);
  for (int g = 0; g < m.ngroup; g++) {
    float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = m.grp_lastdof[g]; i >= 0; i = m.dof_parentid[i]) {
      float vi = WS(v, i);
      for (int k = 0; k < 6; k++) sv[k] += WS(m.w_cdof, i * 6 + k) * vi;
    }
    for (int c = m.grp_start[g]; c < m.grp_start[g] + m.grp_count[g]; c++) {
      float off[3], fr[9], cr[3], vel[3];
      TM_LD(off, WS, m.w_con_off, c * 3, 3);
      TM_LD(fr, WS, m.w_con_frame, c * 9, 9);
      tm_cross(cr, sv, off);
      for (int k = 0; k < 3; k++) vel[k] = sv[3 + k] + cr[k];
      float act = WS(m.w_con_dist, c) < 0.f ? 1.f : 0.f, mu = m.con_mu[c];
      float a0 = tm_dot3(fr, vel), a1 = tm_dot3(fr + 3, vel) * mu, a2 = tm_dot3(fr + 6, vel) * mu;
      int r0 = m.nlim + 4 * c;
      WS(out, r0) = (a0 + a1) * act; WS(out, r0 + 1) = (a0 - a1) * act;
      WS(out, r0 + 2) = (a0 + a2) * act; WS(out, r0 + 3) = (a0 - a2) * act;
    }
  }
}
// out[nv] = J^T f  (contact forces accumulated as one wrench per paw body)
TM_DEV void tm_jtmul(const DModel &m, EnvRef r, int f, int out) {
  for (int i = 0; i < m.nv; i++) WS(out, i) = 0.f;
  for (int l = 0; l < m.nlim; l++) WS(out, m.jnt_dofadr[m.lim_jnt[l]]) += WS(m.w_lim_sign, l) * WS(f, l);
  for (int g = 0; g < m.ngroup; g++) {
    float wr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = m.grp_start[g]; c < m.grp_start[g] + m.grp_count[g]; c++) {
      if (!(WS(m.w_con_dist, c) < 0.f)) continue;
      int r0 = m.nlim + 4 * c;
      float f0 = WS(f, r0), f1 = WS(f, r0 + 1), f2 = WS(f, r0 + 2), f3 = WS(f, r0 + 3), mu = m.con_mu[c];
      float c0 = f0 + f1 + f2 + f3, c1 = mu * (f0 - f1), c2 = mu * (f2 - f3);
      float off[3], fr[9], F[3], T[3];
      TM_LD(off, WS, m.w_con_off, c * 3, 3);
      TM_LD(fr, WS, m.w_con_frame, c * 9, 9);
      for (int k = 0; k < 3; k++) F[k] = c0 * fr[k] + c1 * fr[3 + k] + c2 * fr[6 + k];
      tm_cross(T, off, F);
      for (int k = 0; k < 3; k++) { wr[k] += T[k]; wr[3 + k] += F[k]; }
    }
    for (int i = m.grp_lastdof[g]; i >= 0; i = m.dof_parentid[i]) {
      float s = 0.f;
      for (int k = 0; k < 6; k++) s += WS(m.w_cdof, i * 6 + k) * wr[k];
      WS(out, i) += s;
    }
  }
}

// ------------------------------------------------------------------------------------------ make_constraint
// rows: [limits (jnt order), contacts x 4 pyramid edges]; fills efc_D, efc_aref (needs J*qvel), lim_sign
TM_DEV void tm_make_constraint(const DModel &m, EnvRef r) {
  for (int l = 0; l < m.nlim; l++) {
    int j = m.lim_jnt[l];
    float q = ST(m.s_qpos, m.jnt_qposadr[j]), dmin = q - m.jnt_range[j][0], dmax = m.jnt_range[j][1] - q;
    float pos = fminf(dmin, dmax) - m.jnt_margin[j];
    WS(m.w_lim_sign, l) = pos < 0.f ? (dmin < dmax ? 1.f : -1.f) : 0.f;
    WS(m.w_tmp, l) = pos;  // w_tmp doubles as efc_pos scratch here (nefc rows available)
  }
  // J * qvel for aref: qvel lives in the state buffer, copy to a workspace vector first
  for (int i = 0; i < m.nv; i++) WS(m.w_mv, i) = ST(m.s_qvel, i);
  tm_jmul(m, r, m.w_mv, m.w_efc_jv);
  for (int l = 0; l < m.nlim; l++) {
    int j = m.lim_jnt[l];
    float k, b, imp, pos = WS(m.w_tmp, l);
    tm_kbi(m, m.jnt_solref[j], m.jnt_solimp[j], pos, k, b, imp);
    float R = fmaxf(m.dof_invweight0[m.jnt_dofadr[j]] * (1.f - imp) / imp, TM_MINVAL);
    WS(m.w_efc_D, l) = 1.f / R;
    WS(m.w_efc_aref, l) = -b * WS(m.w_efc_jv, l) - k * imp * pos;
  }
  for (int c = 0; c < m.ncon; c++) {
    float k, b, imp, pos = WS(m.w_con_dist, c);
    tm_kbi(m, m.con_solref[c], m.con_solimp[c], pos, k, b, imp);
    float R = fmaxf(m.con_invweight[c] * (1.f - imp) / imp, TM_MINVAL), D = 1.f / R;
    int r0 = m.nlim + 4 * c;
    for (int e = 0; e < 4; e++) { WS(m.w_efc_D, r0 + e) = D; WS(m.w_efc_aref, r0 + e) = -b * WS(m.w_efc_jv, r0 + e) - k * imp * pos; }
  }
}

// ------------------------------------------------------------------------------------------ fwd_velocity .. fwd_acceleration
// com_vel + rne + passive + actuation -> qfrc_smooth, act_dot, qfrc_actuator; then qacc_smooth = M^-1 qfrc_smooth
TM_DEV void tm_smooth_forces(const DModel &m, EnvRef r) {
  for (int k = 0; k < 6; k++) WS(m.w_cvel, k) = 0.f;
  for (int k = 0; k < 3; k++) { WS(m.w_cacc, k) = 0.f; WS(m.w_cacc, 3 + k) = -m.gravity[k]; }
  for (int b = 1; b < m.nbody; b++) {
    int p = m.body_parentid[b];
    float cv[6], ca[6];
    TM_LD(cv, WS, m.w_cvel, p * 6, 6);
    TM_LD(ca, WS, m.w_cacc, p * 6, 6);
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) {
      int j = m.body_jntadr[b] + jj, da = m.jnt_dofadr[j];
      if (m.jnt_type[j] == 0) {
        for (int rr = 0; rr < 3; rr++) {
          float qv = ST(m.s_qvel, da + rr);
          for (int k = 0; k < 6; k++) { cv[k] += WS(m.w_cdof, (da + rr) * 6 + k) * qv; WS(m.w_cdof_dot, (da + rr) * 6 + k) = 0.f; }
        }
        float cd[3][6], dd[6];
        for (int rr = 0; rr < 3; rr++) {
          TM_LD(cd[rr], WS, m.w_cdof, (da + 3 + rr) * 6, 6);
          tm_motion_cross(dd, cv, cd[rr]);
          TM_SV(WS, m.w_cdof_dot, (da + 3 + rr) * 6, dd, 6);
          float qv = ST(m.s_qvel, da + 3 + rr);
          for (int k = 0; k < 6; k++) ca[k] += dd[k] * qv;
        }
        for (int rr = 0; rr < 3; rr++) { float qv = ST(m.s_qvel, da + 3 + rr); for (int k = 0; k < 6; k++) cv[k] += cd[rr][k] * qv; }
      } else {
        float cd[6], dd[6], qv = ST(m.s_qvel, da);
        TM_LD(cd, WS, m.w_cdof, da * 6, 6);
        tm_motion_cross(dd, cv, cd);
        TM_SV(WS, m.w_cdof_dot, da * 6, dd, 6);
        for (int k = 0; k < 6; k++) { ca[k] += dd[k] * qv; cv[k] += cd[k] * qv; }
      }
    }
    TM_SV(WS, m.w_cvel, b * 6, cv, 6);
    TM_SV(WS, m.w_cacc, b * 6, ca, 6);
  }
  for (int k = 0; k < 6; k++) WS(m.w_cfrc, k) = 0.f;
  for (int b = 1; b < m.nbody; b++) {
    float I[10], cv[6], ca[6], f1[6], t[6], f2[6];
    TM_LD(I, WS, m.w_cinert, b * 10, 10);
    TM_LD(cv, WS, m.w_cvel, b * 6, 6);
    TM_LD(ca, WS, m.w_cacc, b * 6, 6);
    tm_inert_mul(f1, I, ca);
    tm_inert_mul(t, I, cv);
    tm_motion_cross_force(f2, cv, t);
    for (int k = 0; k < 6; k++) WS(m.w_cfrc, b * 6 + k) = f1[k] + f2[k];
  }
  for (int b = m.nbody - 1; b >= 1; b--) {
    int p = m.body_parentid[b];
    if (p == 0) continue;
    for (int k = 0; k < 6; k++) WS(m.w_cfrc, p * 6 + k) += WS(m.w_cfrc, b * 6 + k);
  }
  // actuation (filter dynamics, fixed gain, no bias: torque-actuator model)
  for (int i = 0; i < m.nv; i++) ST(m.s_qfrc_actuator, i) = 0.f;
  for (int a = 0; a < m.nu; a++) {
    float ctrl = fminf(fmaxf(WS(m.w_ctrl, a), m.act_ctrlrange[a][0]), m.act_ctrlrange[a][1]);
    float act = ST(m.s_act, a);
    WS(m.w_act_dot, a) = (ctrl - act) / fmaxf(TM_MINVAL, m.act_tau[a]);
    float force = m.act_gain[a] * act;
    for (int e = m.act_madr[a]; e < m.act_madr[a + 1]; e++) ST(m.s_qfrc_actuator, m.act_mdof[e]) += m.act_mval[e] * force;
  }
  // qfrc_smooth = passive - bias + actuator
  for (int i = 0; i < m.nv; i++) {
    int b = m.dof_bodyid[i];
    float bias = 0.f;
    for (int k = 0; k < 6; k++) bias += WS(m.w_cdof, i * 6 + k) * WS(m.w_cfrc, b * 6 + k);
    WS(m.w_qfrc_smooth, i) = -m.dof_damping[i] * ST(m.s_qvel, i) - bias + ST(m.s_qfrc_actuator, i);
  }
  for (int j = 0; j < m.njnt; j++) {
    if (m.jnt_type[j] != 3 || m.jnt_stiffness[j] == 0.f) continue;
    int qa = m.jnt_qposadr[j];
    WS(m.w_qfrc_smooth, m.jnt_dofadr[j]) += -m.jnt_stiffness[j] * (ST(m.s_qpos, qa) - m.qpos_spring[qa]);
  }
  for (int i = 0; i < m.nv; i++) WS(m.w_qacc_smooth, i) = WS(m.w_qfrc_smooth, i);
  tm_solve(m, r, m.w_qacc_smooth);
}

// ------------------------------------------------------------------------------------------ solver.solve (CG)
struct TmLS { float alpha, cost, d0, d1; };
TM_DEV float tm_dotw(EnvRef r, int a, int b, int n) { float s = 0.f; for (int i = 0; i < n; i++) s += WS(a, i) * WS(b, i); return s; }

// cost at qacc (workspace vector q): fills Ma (w_Ma), Jaref (w_efc_Jaref); returns cost, writes gauss
TM_DEV float tm_eval_cost(const DModel &m, EnvRef r, int q, float &gauss) {
  tm_mul_m(m, r, q, m.w_Ma);
  tm_jmul(m, r, q, m.w_efc_Jaref);
  float cost = 0.f;
  for (int e = 0; e < m.nefc; e++) {
    float ja = WS(m.w_efc_Jaref, e) - WS(m.w_efc_aref, e);
    WS(m.w_efc_Jaref, e) = ja;
    if (ja < 0.f) cost += WS(m.w_efc_D, e) * ja * ja;
  }
  float g = 0.f;
  for (int i = 0; i < m.nv; i++) g += (WS(m.w_Ma, i) - WS(m.w_qfrc_smooth, i)) * (WS(q, i) - WS(m.w_qacc_smooth, i));
  gauss = 0.5f * g;
  return 0.5f * cost + gauss;
}
// _update_constraint given Jaref, Ma, qacc: efc_force, qfrc_constraint, gauss, cost
TM_DEV float tm_update_constraint(const DModel &m, EnvRef r, float &gauss) {
  float cost = 0.f;
  for (int e = 0; e < m.nefc; e++) {
    float ja = WS(m.w_efc_Jaref, e), D = WS(m.w_efc_D, e);
    bool act = ja < 0.f;
    WS(m.w_efc_force, e) = act ? D * -ja : 0.f;
    if (act) cost += D * ja * ja;
  }
  tm_jtmul(m, r, m.w_efc_force, m.w_qfrc_constraint);
  float g = 0.f;
  for (int i = 0; i < m.nv; i++) g += (WS(m.w_Ma, i) - WS(m.w_qfrc_smooth, i)) * (WS(m.w_qacc, i) - WS(m.w_qacc_smooth, i));
  gauss = 0.5f * g;
  return 0.5f * cost + gauss;
}
TM_DEV void tm_update_gradient(const DModel &m, EnvRef r) {
  for (int i = 0; i < m.nv; i++) {
    float g = WS(m.w_Ma, i) - WS(m.w_qfrc_smooth, i) - WS(m.w_qfrc_constraint, i);
    WS(m.w_grad, i) = g; WS(m.w_Mgrad, i) = g;
  }
  tm_solve(m, r, m.w_Mgrad);
}
TM_DEV TmLS tm_ls_point(const DModel &m, EnvRef r, float alpha, float g0, float g1, float g2) {
  float q0 = g0, q1 = g1, q2 = g2;
  for (int e = 0; e < m.nefc; e++) {
    float ja = WS(m.w_efc_Jaref, e), jv = WS(m.w_efc_jv, e);
    if (ja + alpha * jv < 0.f) {
      float D = WS(m.w_efc_D, e);
      q0 += 0.5f * ja * ja * D; q1 += jv * ja * D; q2 += 0.5f * jv * jv * D;
    }
  }
  TmLS p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.d0 = 2.f * alpha * q2 + q1;
  p.d1 = 2.f * q2 + (q2 == 0.f ? TM_MINVAL : 0.f);
  return p;
}
TM_DEV void tm_linesearch(const DModel &m, EnvRef r, float gauss) {
  float scale = m.meaninertia * (float)(m.nv > 1 ? m.nv : 1);
  float smag = sqrtf(tm_dotw(r, m.w_search, m.w_search, m.nv)) * scale;
  float gtol = m.tolerance * m.ls_tolerance * smag;
  tm_mul_m(m, r, m.w_search, m.w_mv);
  tm_jmul(m, r, m.w_search, m.w_efc_jv);
  float g0 = gauss, g1 = tm_dotw(r, m.w_search, m.w_Ma, m.nv) - tm_dotw(r, m.w_search, m.w_qfrc_smooth, m.nv),
        g2 = 0.5f * tm_dotw(r, m.w_search, m.w_mv, m.nv);
  TmLS p0 = tm_ls_point(m, r, 0.f, g0, g1, g2);
  TmLS lo0 = tm_ls_point(m, r, p0.alpha - p0.d0 / p0.d1, g0, g1, g2);
  bool lesser = lo0.d0 < p0.d0;
  TmLS hi = lesser ? p0 : lo0, lo = lesser ? lo0 : p0;
  bool swap = true;
  for (int it = 0; it < m.ls_iterations; it++) {
    bool done = !swap || ((lo.d0 < 0.f) && (lo.d0 > -gtol)) || ((hi.d0 > 0.f) && (hi.d0 < gtol));
    if (done) break;
    TmLS lo_next = tm_ls_point(m, r, lo.alpha - lo.d0 / lo.d1, g0, g1, g2);
    TmLS hi_next = tm_ls_point(m, r, hi.alpha - hi.d0 / hi.d1, g0, g1, g2);
    TmLS mid = tm_ls_point(m, r, 0.5f * (lo.alpha + hi.alpha), g0, g1, g2);
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
  bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  float alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  float ia = improved ? alpha : 0.f;
  for (int i = 0; i < m.nv; i++) { WS(m.w_qacc, i) += WS(m.w_search, i) * ia; WS(m.w_Ma, i) += WS(m.w_mv, i) * ia; }
  for (int e = 0; e < m.nefc; e++) WS(m.w_efc_Jaref, e) += WS(m.w_efc_jv, e) * ia;
}
TM_DEV void tm_solve_cg(const DModel &m, EnvRef r) {
  float gauss, scale = m.meaninertia * (float)(m.nv > 1 ? m.nv : 1);
  // warm start: the cheaper of qacc_warmstart and qacc_smooth
  for (int i = 0; i < m.nv; i++) WS(m.w_qacc, i) = ST(m.s_warm, i);
  float cw = tm_eval_cost(m, r, m.w_qacc, gauss);
  float cs = tm_eval_cost(m, r, m.w_qacc_smooth, gauss);
  if (cw < cs) cs = tm_eval_cost(m, r, m.w_qacc, gauss);
  else for (int i = 0; i < m.nv; i++) WS(m.w_qacc, i) = WS(m.w_qacc_smooth, i);
  float cost = tm_update_constraint(m, r, gauss), prev_cost = INFINITY;
  tm_update_gradient(m, r);
  for (int i = 0; i < m.nv; i++) WS(m.w_search, i) = -WS(m.w_Mgrad, i);
  for (int it = 0; it < m.iterations; it++) {
    if (m.iterations != 1) {
      float improvement = (prev_cost - cost) / scale;
      float gradient = sqrtf(tm_dotw(r, m.w_grad, m.w_grad, m.nv)) / scale;
      if (improvement < m.tolerance || gradient < m.tolerance) break;
    }
    tm_linesearch(m, r, gauss);
    float den = tm_dotw(r, m.w_grad, m.w_Mgrad, m.nv);
    for (int i = 0; i < m.nv; i++) WS(m.w_tmp, i) = WS(m.w_Mgrad, i);  // previous Mgrad
    prev_cost = cost;
    cost = tm_update_constraint(m, r, gauss);
    tm_update_gradient(m, r);
    float num = 0.f;
    for (int i = 0; i < m.nv; i++) num += WS(m.w_grad, i) * (WS(m.w_Mgrad, i) - WS(m.w_tmp, i));
    float beta = fmaxf(0.f, num / fmaxf(TM_MINVAL, den));
    for (int i = 0; i < m.nv; i++) WS(m.w_search, i) = -WS(m.w_Mgrad, i) + beta * WS(m.w_search, i);
  }
  for (int i = 0; i < m.nv; i++) ST(m.s_warm, i) = WS(m.w_qacc, i);
}

// ------------------------------------------------------------------------------------------ forward / euler
TM_DEV void tm_forward(const DModel &m, EnvRef r) {
  tm_kinematics_com(m, r);
  tm_crb(m, r);
  tm_factor(m, r, 0.f);
  tm_collision(m, r);
  tm_make_constraint(m, r);
  tm_smooth_forces(m, r);
  tm_solve_cg(m, r);
}
// forward.euler + _advance: implicit-in-damping velocity update, semi-implicit position update
TM_DEV void tm_euler(const DModel &m, EnvRef r) {
  float h = m.timestep;
  tm_factor(m, r, h);
  for (int i = 0; i < m.nv; i++) WS(m.w_tmp, i) = WS(m.w_qfrc_smooth, i) + WS(m.w_qfrc_constraint, i);
  tm_solve(m, r, m.w_tmp);
  for (int a = 0; a < m.nu; a++) ST(m.s_act, a) += WS(m.w_act_dot, a) * h;
  for (int i = 0; i < m.nv; i++) ST(m.s_qvel, i) += WS(m.w_tmp, i) * h;
  for (int j = 0; j < m.njnt; j++) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == 0) {
      for (int k = 0; k < 3; k++) ST(m.s_qpos, qa + k) += h * ST(m.s_qvel, da + k);
      float v[3], q[4], qr[4], q2[4];
      TM_LD(v, ST, m.s_qvel, da + 3, 3);
      TM_LD(q, ST, m.s_qpos, qa + 3, 4);
      float nn = tm_normalize3(v), ang = h * nn * 0.5f, sn = sinf(ang), cs = cosf(ang);
      qr[0] = cs; qr[1] = v[0] * sn; qr[2] = v[1] * sn; qr[3] = v[2] * sn;
      tm_quat_mul(q2, q, qr);
      tm_normalize4(q2);
      TM_SV(ST, m.s_qpos, qa + 3, q2, 4);
    } else {
      ST(m.s_qpos, qa) += h * ST(m.s_qvel, da);
    }
  }
  ST(m.s_time, 0) += h;
}
