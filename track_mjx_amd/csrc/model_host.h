// csrc/model_host.h — host side: model blob -> DModel (constants, sparse tables, buffer layouts).
// Plain C++ (no HIP) so that the test-only host emulation of the kernel bodies can share it.
#pragma once
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "dmodel.h"
#include "wave_layout.h"

namespace tmjx_host {

// does the dof tree equal the compile-time chain table of the rodent (wave_layout.h)?  Required by the register-resident
// factorisation: dofs 0..TRUNK-1 one chain from the root, every leaf chain X(first, n, d0) = consecutive dofs hanging off
// trunk dof d0-1, M rows stored back to back in dof order.
inline bool rodent_chains_match(const DModel &m) {
  constexpr WLayout ks(TMW_RODENT_DIMS);
  if (m.nv != ks.nv || m.nnz != ks.nnz) return false;
  std::vector<int> parent(m.nv, -2);
  for (int i = 0; i < TMW_RODENT_TRUNK; i++) parent[i] = i - 1;
#define TMW_X(first, n, d0) for (int k = 0; k < n; k++) parent[first + k] = k ? first + k - 1 : d0 - 1;
  TMW_RODENT_LEAF_CHAINS(TMW_X)
#undef TMW_X
  int adr = 0;
  for (int i = 0; i < m.nv; i++) {
    if (parent[i] != m.dof_parentid[i] || m.dof_Madr[i] != adr) return false;
    if (m.dof_depth[i] != (parent[i] < 0 ? 0 : m.dof_depth[parent[i]] + 1)) return false;
    adr += m.dof_depth[i] + 1;
  }
  // the lean LDS map stores ONE contact normal (wave_layout.h): every contact slot must be against the same plane geom of a body that does
  // not move (outside the walker's tree)
  for (int cc = 0; cc < m.ncon; cc++) {
    if (m.con_body1[cc] != m.con_body1[0] || m.body_moving[m.con_body1[cc]]) return false;
    for (int k = 0; k < 3; k++) if (m.con_g1_pos[cc][k] != m.con_g1_pos[0][k]) return false;
    for (int k = 0; k < 4; k++) if (m.con_g1_quat[cc][k] != m.con_g1_quat[0][k]) return false;
    if (m.con_mu[cc] != m.con_mu[0]) return false;      // ... and reads ONE friction coefficient (TMW_MU)
  }
  // ... keeps the per-dof table as one packed word per dof in registers (dmodel.h: tpack) and takes the limit row of dof i as i - 6 (every hinge
  // of the rodent is limited, rows in dof order; the free joint's six dofs have none)
  for (int i = 0; i < m.nv; i++) {
    if (m.tpack[i] == 0 && i > 0) return false;
    if (m.dof_limrow[i] != (i >= 6 ? i - 6 : -1)) return false;
  }
  if (m.nv > 128 || m.nlim != m.nv - 6) return false;
  // ... and runs the kinematics with ONE body slot (wave_physics.h: tmw_position, "head-4 body tree"): body 0 the world at the origin, body 1 a
  // jointless child of the world, body 2 the free-joint root (a child of the world: absolute by its qpos), body 3 welded to body 2, at most 64
  // bodies behind them and no other free joint
  if (m.nbody < 5 || m.nbody - 4 > 64 || m.nround_body < 1) return false;
  if (m.body_jntnum[0] || m.body_jntnum[1] || m.body_jntnum[2] != 1 || m.body_jntnum[3]) return false;
  if (m.jnt_type[m.body_jntadr[2]] != 0 || m.body_parentid[1] != 0 || m.body_parentid[2] != 0 || m.body_parentid[3] != 2) return false;
  if (m.scan_parent[0] != -1 || m.scan_parent[1] != -1 || m.scan_parent[2] != -1 || m.scan_parent[3] != 2) return false;
  for (int b = 4; b < m.nbody; b++)
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) if (m.jnt_type[m.body_jntadr[b] + jj] == 0) return false;
  return adr == m.nnz;
}
inline WLayout make_wave_layout(const DModel &m, bool allow_chains = true) {
  return WLayout(m.nbody, m.njnt, m.nq, m.nv, m.nu, m.ncon, m.nlim, m.nnz, m.ngroup, m.nround_body, m.nround_dof,
                 allow_chains && rodent_chains_match(m) ? 1 : 0);
}

struct BlobEntry { int code = -1, count = 0; const unsigned char *data = nullptr; };

inline bool blob_find(const void *blob, size_t n, const char *name, BlobEntry &out) {
  const unsigned char *p = (const unsigned char *)blob;
  if (n < 16) return false;
  uint32_t magic, ver, ne;
  memcpy(&magic, p, 4); memcpy(&ver, p + 4, 4); memcpy(&ne, p + 8, 4);
  if (magic != 0x584A4D54u || ver != 1) return false;
  size_t off = 16;
  for (uint32_t i = 0; i < ne && off + 40 <= n; i++) {
    int32_t code, count;
    memcpy(&code, p + off + 32, 4); memcpy(&count, p + off + 36, 4);
    size_t nb = (size_t)count * (code == 0 ? 4 : 8), padded = nb + ((8 - nb % 8) % 8);
    if (off + 40 + nb > n) return false;
    if (strncmp((const char *)(p + off), name, 32) == 0) { out.code = code; out.count = count; out.data = p + off + 40; return true; }
    off += 40 + padded;
  }
  return false;
}

struct Reader {
  const void *blob; size_t n; std::string err;
  bool ints(const char *name, int *dst, int expect, int cap = -1) {
    BlobEntry e;
    if (!blob_find(blob, n, name, e) || e.code != 0) { err = std::string("blob entry missing: ") + name; return false; }
    if ((expect >= 0 && e.count != expect) || (cap >= 0 && e.count > cap)) { err = std::string("blob entry has wrong size: ") + name; return false; }
    for (int i = 0; i < e.count; i++) { int32_t v; memcpy(&v, e.data + 4 * i, 4); dst[i] = v; }
    return true;
  }
  int count(const char *name) { BlobEntry e; return blob_find(blob, n, name, e) ? e.count : -1; }
  bool floats(const char *name, float *dst, int expect) {
    BlobEntry e;
    if (!blob_find(blob, n, name, e) || e.code != 1) { err = std::string("blob entry missing: ") + name; return false; }
    if (e.count != expect) { err = std::string("blob entry has wrong size: ") + name; return false; }
    for (int i = 0; i < e.count; i++) { double v; memcpy(&v, e.data + 8 * i, 8); dst[i] = (float)v; }
    return true;
  }
};

// Fills every field of DModel except the clip pointers. Returns false and sets `err` on malformed input.
inline bool build_dmodel(const void *blob, size_t nbytes, DModel &m, std::string &err) {
  memset(&m, 0, sizeof(DModel));
  Reader R{blob, nbytes, ""};
  int dims[6];
  if (!R.ints("dims", dims, 6)) { err = R.err; return false; }
  m.nbody = dims[0]; m.njnt = dims[1]; m.nq = dims[2]; m.nv = dims[3]; m.nu = dims[4]; m.ncon = dims[5];
  if (m.nbody > TM_MAXB || m.nv > TM_MAXV || m.nq > TM_MAXQ || m.nu > TM_MAXU || m.ncon > TM_MAXC || m.njnt > TM_MAXV) {
    err = "model exceeds the compiled-in maximum dimensions"; return false;
  }
  bool ok = true;
  std::vector<int> rootid(m.nbody), jlim(m.njnt), dof_jnt(m.nv);
  ok = ok && R.ints("body_parentid", m.body_parentid, m.nbody) && R.ints("body_rootid", rootid.data(), m.nbody) &&
       R.ints("body_jntadr", m.body_jntadr, m.nbody) && R.ints("body_jntnum", m.body_jntnum, m.nbody) &&
       R.ints("body_dofadr", m.body_dofadr, m.nbody) && R.ints("body_dofnum", m.body_dofnum, m.nbody) &&
       R.ints("jnt_type", m.jnt_type, m.njnt) && R.ints("jnt_bodyid", m.jnt_bodyid, m.njnt) &&
       R.ints("jnt_qposadr", m.jnt_qposadr, m.njnt) && R.ints("jnt_dofadr", m.jnt_dofadr, m.njnt) &&
       R.ints("jnt_limited", jlim.data(), m.njnt) && R.ints("dof_bodyid", m.dof_bodyid, m.nv) &&
       R.ints("dof_parentid", m.dof_parentid, m.nv);
  ok = ok && R.floats("body_pos", &m.body_pos[0][0], m.nbody * 3) && R.floats("body_quat", &m.body_quat[0][0], m.nbody * 4) &&
       R.floats("body_mass", m.body_mass, m.nbody) && R.floats("body_ipos", &m.body_ipos[0][0], m.nbody * 3) &&
       R.floats("body_iquat", &m.body_iquat[0][0], m.nbody * 4) && R.floats("body_inertia", &m.body_inertia[0][0], m.nbody * 3) &&
       R.floats("jnt_pos", &m.jnt_pos[0][0], m.njnt * 3) && R.floats("jnt_axis", &m.jnt_axis[0][0], m.njnt * 3) &&
       R.floats("jnt_range", &m.jnt_range[0][0], m.njnt * 2) && R.floats("jnt_stiffness", m.jnt_stiffness, m.njnt) &&
       R.floats("jnt_solref", &m.jnt_solref[0][0], m.njnt * 2) && R.floats("jnt_solimp", &m.jnt_solimp[0][0], m.njnt * 5) &&
       R.floats("jnt_margin", m.jnt_margin, m.njnt) && R.floats("qpos0", m.qpos0, m.nq) &&
       R.floats("qpos_spring", m.qpos_spring, m.nq) && R.floats("dof_damping", m.dof_damping, m.nv) &&
       R.floats("dof_armature", m.dof_armature, m.nv) && R.floats("dof_invweight0", m.dof_invweight0, m.nv) &&
       R.floats("act_gain", m.act_gain, m.nu) && R.floats("act_tau", m.act_tau, m.nu) &&
       R.floats("act_ctrlrange", &m.act_ctrlrange[0][0], m.nu * 2) && R.floats("gravity", m.gravity, 3) &&
       R.floats("meaninertia", &m.meaninertia, 1);
  if (!ok) { err = R.err; return false; }
  for (int j = 0; j < m.njnt; j++)
    if (m.jnt_type[j] != 0 && m.jnt_type[j] != 3) { err = "only free and hinge joints are supported on this path"; return false; }
  // sparse actuator moments
  std::vector<float> mom((size_t)m.nu * m.nv);
  if (!R.floats("act_moment", mom.data(), m.nu * m.nv)) { err = R.err; return false; }
  int ne = 0;
  for (int a = 0; a < m.nu; a++) {
    m.act_madr[a] = ne;
    for (int i = 0; i < m.nv; i++) if (mom[(size_t)a * m.nv + i] != 0.f) {
      if (ne >= 128) { err = "actuator moment has too many non-zeros"; return false; }
      m.act_mdof[ne] = i; m.act_mval[ne] = mom[(size_t)a * m.nv + i]; ne++;
    }
  }
  m.act_madr[m.nu] = ne;
  // single moving tree
  m.root_body = -1;
  for (int b = 1; b < m.nbody; b++) {
    bool moving = false;
    for (int c = b; c > 0; c = m.body_parentid[c]) if (m.body_dofnum[c] > 0) moving = true;
    if (moving) { if (m.root_body < 0) m.root_body = rootid[b]; if (rootid[b] != m.root_body) { err = "more than one moving tree"; return false; } }
    m.body_moving[b] = rootid[b] == m.root_body && m.root_body >= 0;
    if (!m.body_moving[b] && m.body_mass[b] != 0.f) { err = "static bodies must be massless"; return false; }
  }
  // tree-sparse M layout: row i = [M(i,i), M(i,parent), M(i,grandparent), ...]
  int nnz = 0;
  for (int i = 0; i < m.nv; i++) {
    int d = 0;
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j]) d++;
    m.dof_depth[i] = d; m.dof_Madr[i] = nnz; nnz += d + 1;
  }
  m.nnz = nnz;
  m.nlim = 0;
  for (int j = 0; j < m.njnt; j++) if (jlim[j] && m.jnt_type[j] == 3) m.lim_jnt[m.nlim++] = j;
  m.nefc = m.nlim + 4 * m.ncon;
  // contacts
  std::vector<float> fr(m.ncon * 3), binv(m.nbody * 2);
  ok = R.ints("con_type", m.con_type, m.ncon) && R.ints("con_sub", m.con_sub, m.ncon) && R.ints("con_body1", m.con_body1, m.ncon) &&
       R.ints("con_body2", m.con_body2, m.ncon) && R.floats("con_friction", fr.data(), m.ncon * 3) &&
       R.floats("con_solref", &m.con_solref[0][0], m.ncon * 2) && R.floats("con_solimp", &m.con_solimp[0][0], m.ncon * 5) &&
       R.floats("con_g1_pos", &m.con_g1_pos[0][0], m.ncon * 3) && R.floats("con_g1_quat", &m.con_g1_quat[0][0], m.ncon * 4) &&
       R.floats("con_g2_pos", &m.con_g2_pos[0][0], m.ncon * 3) && R.floats("con_g2_quat", &m.con_g2_quat[0][0], m.ncon * 4) &&
       R.floats("con_g2_size", &m.con_g2_size[0][0], m.ncon * 3) && R.floats("body_invweight0", binv.data(), m.nbody * 2);
  if (!ok) { err = R.err; return false; }
  float optf[4]; int opti[3], envi[7];
  ok = R.floats("opt_f", optf, 4) && R.ints("opt_i", opti, 3) && R.ints("env_i", envi, 7) && R.floats("reward_f", m.rw, 25);
  if (!ok) { err = R.err; return false; }
  m.timestep = optf[0]; m.tolerance = optf[1]; m.ls_tolerance = optf[2]; m.impratio = optf[3];
  m.iterations = opti[0]; m.ls_iterations = opti[1]; m.n_frames = opti[2];
  m.mocap_hz = envi[0]; m.clip_length = envi[1]; m.traj_length = envi[2]; m.window = envi[3]; m.torso_idx = envi[4]; m.episode_length = envi[5]; m.auto_reset = envi[6];
  m.ngroup = 0;
  for (int c = 0; c < m.ncon; c++) {
    if (m.body_moving[m.con_body1[c]] || !m.body_moving[m.con_body2[c]]) { err = "contact slots must be (static geom, moving geom)"; return false; }
    float mu = fr[c * 3];
    m.con_mu[c] = mu;
    float t = binv[m.con_body1[c] * 2] + binv[m.con_body2[c] * 2];
    m.con_invweight[c] = (t + mu * mu * t) * 2.f * mu * mu / m.impratio;  // pyramidal edge, condim 3
    int b = m.con_body2[c];
    if (m.ngroup == 0 || m.grp_body[m.ngroup - 1] != b) {
      if (m.ngroup >= TM_MAXG) {  // a body may re-appear non-contiguously: start a new group anyway
        err = "too many contact groups"; return false;
      }
      int g = m.ngroup++;
      m.grp_body[g] = b; m.grp_start[g] = c; m.grp_count[g] = 0;
      int last = -1;
      for (int bb = b; bb > 0 && last < 0; bb = m.body_parentid[bb]) if (m.body_dofnum[bb]) last = m.body_dofadr[bb] + m.body_dofnum[bb] - 1;
      m.grp_lastdof[g] = last;
    }
    m.grp_count[m.ngroup - 1]++;
  }
  m.n_joint_idx = R.count("joint_idxs"); m.n_body_idx = R.count("body_idxs"); m.n_endeff_idx = R.count("endeff_idxs");
  if (m.n_joint_idx <= 0 || m.n_body_idx <= 0 || m.n_endeff_idx <= 0 || m.n_joint_idx > TM_MAXIDX || m.n_body_idx > TM_MAXIDX || m.n_endeff_idx > TM_MAXIDX) {
    err = "bad tracked index lists"; return false;
  }
  R.ints("joint_idxs", m.joint_idxs, -1); R.ints("body_idxs", m.body_idxs, -1); R.ints("endeff_idxs", m.endeff_idxs, -1);
  for (int k = 0; k < m.n_endeff_idx; k++) if (m.endeff_idxs[k] < 0 || m.endeff_idxs[k] >= m.nbody) { err = "end effector id out of range"; return false; }
  if (m.torso_idx < 0 || m.torso_idx >= m.nbody || m.window < 3 || m.traj_length < 1) { err = "bad task configuration"; return false; }
  int T = m.traj_length;
  m.ref_obs_size = T * 3 + T * 4 + T * m.n_joint_idx + T * m.n_body_idx * 3;
  m.obs_size = m.ref_obs_size + (m.nq - 7) + (m.nv - 6) + m.nv + 1 + 3 + 3 * m.n_endeff_idx;
  // ---- layouts
  int s = 0;
  m.s_qpos = s; s += m.nq; m.s_qvel = s; s += m.nv; m.s_act = s; s += m.nu; m.s_warm = s; s += m.nv; m.s_time = s; s += 1;
  m.nphys = s;
  m.s_xpos = s; s += m.nbody * 3; m.s_xmat_torso = s; s += 9; m.s_qfrc_actuator = s; s += m.nv;
  m.s_prev_ctrl = s; s += m.nu; m.s_action_buffer = s; s += m.window * m.nu; m.s_done = s; s += 1; m.s_steps = s; s += 1;
  m.s_first_phys = s; s += m.nphys; m.s_first_obs = s; s += m.obs_size; m.s_first_prev_ctrl = s; s += m.nu;
  m.s_rows = s;
  m.i_clip_idx = 0; m.i_start_frame = 1; m.i_buffer_index = 2; m.i_nan_count = 3; m.i_rows = 4;
  int w = 0;
#define WROW(f, cnt) m.f = w; w += (cnt)
  WROW(w_ctrl, m.nu); WROW(w_xquat, m.nbody * 4); WROW(w_xanchor, m.njnt * 3); WROW(w_xaxis, m.njnt * 3); WROW(w_xipos, m.nbody * 3);
  WROW(w_cinert, m.nbody * 10); WROW(w_cdof, m.nv * 6); WROW(w_crb, m.nbody * 10); WROW(w_M, m.nnz); WROW(w_LD, m.nnz); WROW(w_Dinv, m.nv);
  WROW(w_cvel, m.nbody * 6); WROW(w_cdof_dot, m.nv * 6); WROW(w_cacc, m.nbody * 6); WROW(w_cfrc, m.nbody * 6);
  WROW(w_qfrc_smooth, m.nv); WROW(w_qacc_smooth, m.nv); WROW(w_act_dot, m.nu);
  WROW(w_con_dist, m.ncon); WROW(w_con_off, m.ncon * 3); WROW(w_con_frame, m.ncon * 9);
  WROW(w_efc_D, m.nefc); WROW(w_efc_aref, m.nefc); WROW(w_efc_Jaref, m.nefc); WROW(w_efc_jv, m.nefc); WROW(w_lim_sign, m.nlim);
  WROW(w_qacc, m.nv); WROW(w_Ma, m.nv); WROW(w_grad, m.nv); WROW(w_Mgrad, m.nv); WROW(w_search, m.nv); WROW(w_mv, m.nv);
  WROW(w_qfrc_constraint, m.nv); WROW(w_tmp, m.nefc > m.nv ? m.nefc : m.nv); WROW(w_efc_force, m.nefc); WROW(w_com, 3); WROW(w_solver_stats, 4); WROW(w_efc_in, m.nefc);
#undef WROW
  m.w_rows = w;

  // ================= wave-per-env kernel tables =================
  if (m.nnz > 1280 || m.nbody > 128 || m.nv > 128) { err = "model too large for the wave kernel tables"; return false; }
  std::vector<int> blevel(m.nbody, 0);
  int maxlevel = 0;
  for (int b = 1; b < m.nbody; b++) { blevel[b] = blevel[m.body_parentid[b]] + 1; if (blevel[b] > maxlevel) maxlevel = blevel[b]; }
  for (int b = 0; b < m.nbody; b++) {
    bool has_free = false;
    for (int jj = 0; jj < m.body_jntnum[b]; jj++) if (m.jnt_type[m.body_jntadr[b] + jj] == 0) has_free = true;
    m.scan_parent[b] = (b == 0 || has_free || m.body_parentid[b] == 0) ? -1 : m.body_parentid[b];
    int last = -1;
    for (int c = b; c > 0 && last < 0; c = m.body_parentid[c]) if (m.body_dofnum[c]) last = m.body_dofadr[c] + m.body_dofnum[c] - 1;
    m.body_lastdof[b] = last;
  }
  for (int b = 0; b < m.nbody; b++) {      // flat kinematics record of every body (dmodel.h: body_kin)
    float *r = m.body_kin[b];
    for (int k = 0; k < 16; k++) r[k] = 0.f;
    for (int k = 0; k < 3; k++) r[k] = m.body_pos[b][k];
    for (int k = 0; k < 4; k++) r[3 + k] = m.body_quat[b][k];
    int jt = -1, qa = 0;
    if (m.body_jntnum[b] > 0) {
      const int j = m.body_jntadr[b];
      jt = m.jnt_type[j]; qa = m.jnt_qposadr[j];
      for (int k = 0; k < 3; k++) { r[9 + k] = m.jnt_pos[j][k]; r[12 + k] = m.jnt_axis[j][k]; }
      r[15] = m.qpos0[qa];
    }
    memcpy(&r[7], &jt, 4); memcpy(&r[8], &qa, 4);
  }
  {  // children lists (body ids ascending)
    int a = 0;
    for (int p = 0; p < m.nbody; p++) {
      m.child_adr[p] = a;
      for (int c = 1; c < m.nbody; c++) if (m.body_parentid[c] == p && c != p) m.child_ids[a++] = c;
    }
    m.child_adr[m.nbody] = a;
    // up-sweep levels: deepest parents first; world (level 0) is never a target
    int nl = 0, la = 0;
    for (int L = maxlevel - 1; L >= 1; L--) {
      m.lvl_adr[nl] = la;
      for (int p = 1; p < m.nbody; p++) if (blevel[p] == L && m.child_adr[p + 1] > m.child_adr[p]) m.lvl_parents[la++] = p;
      if (la > m.lvl_adr[nl]) nl++;
    }
    m.lvl_adr[nl] = la;
    m.nlevel = nl;
  }
  int scan_depth = 0;
  for (int b = 0; b < m.nbody; b++) { int d = 0; for (int c = b; m.scan_parent[c] >= 0; c = m.scan_parent[c]) d++; if (d > scan_depth) scan_depth = d; }
  m.nround_body = 0; while ((1 << m.nround_body) < scan_depth + 1) m.nround_body++;
  int maxdd = 0;
  for (int i = 0; i < m.nv; i++) if (m.dof_depth[i] > maxdd) maxdd = m.dof_depth[i];
  m.nround_dof = 0; while ((1 << m.nround_dof) < maxdd + 1) m.nround_dof++;
  for (int j = 0; j < m.njnt; j++) {
    int nd = m.jnt_type[j] == 0 ? 6 : 1;
    for (int k = 0; k < nd; k++) {
      int i = m.jnt_dofadr[j] + k;
      m.dof_jntid[i] = j;
      m.dof_freetrans[i] = (m.jnt_type[j] == 0 && k < 3) ? 1 : 0;
      // velocity prefix seen by cdof_dot: free rotational dofs all use the prefix after the 3 translations
      m.dof_vpar[i] = (m.jnt_type[j] == 0 && k >= 3) ? m.jnt_dofadr[j] + 2 : m.dof_parentid[i];
      m.dof_stiffness[i] = m.jnt_type[j] == 3 ? m.jnt_stiffness[j] : 0.f;
      m.dof_qposadr[i] = m.jnt_type[j] == 3 ? m.jnt_qposadr[j] : 0;
      m.dof_qspring[i] = m.jnt_type[j] == 3 ? m.qpos_spring[m.jnt_qposadr[j]] : 0.f;
      m.dof_limrow[i] = -1;
    }
  }
  for (int l = 0; l < m.nlim; l++) m.dof_limrow[m.jnt_dofadr[m.lim_jnt[l]]] = l;
  for (int i = 0; i < m.nv; i++) {  // descendants are a contiguous dof range (depth-first numbering)
    int nd = 0;
    for (int k = i + 1; k < m.nv; k++) { bool desc = false; for (int a = m.dof_parentid[k]; a >= 0; a = m.dof_parentid[a]) if (a == i) desc = true; if (desc) nd++; else break; }
    m.dof_ndesc[i] = nd;
    int total = 0;
    for (int k = i + 1; k < m.nv; k++) for (int a = m.dof_parentid[k]; a >= 0; a = m.dof_parentid[a]) if (a == i) total++;
    if (total != nd) { err = "dof numbering is not depth-first"; return false; }
    int k = 0;
    for (int j = i; j >= 0; j = m.dof_parentid[j], k++) { m.anc_dof[m.dof_Madr[i] + k] = (uint8_t)j; m.anc_Madr[m.dof_Madr[i] + k] = (uint16_t)m.dof_Madr[j]; }
  }
  {  // column access table: M(k,i) for every descendant k of i, k ascending
    int a = 0;
    for (int i = 0; i < m.nv; i++) {
      m.dof_coladr[i] = a;
      for (int k = i + 1; k <= i + m.dof_ndesc[i]; k++) m.col_off[a++] = (uint16_t)(m.dof_Madr[k] + m.dof_depth[k] - m.dof_depth[i]);
    }
    m.dof_coladr[m.nv] = a;
    m.total_mass = 0.f;
    for (int b = 0; b < m.nbody; b++) if (m.body_moving[b]) m.total_mass += m.body_mass[b];
  }
  {  // per-dof actuator gather lists (transpose of the sparse moment)
    int a = 0;
    for (int i = 0; i < m.nv; i++) {
      m.dof_act_adr[i] = a;
      for (int u = 0; u < m.nu; u++) for (int e = m.act_madr[u]; e < m.act_madr[u + 1]; e++) if (m.act_mdof[e] == i) {
        if (a >= 128) { err = "too many actuator couplings"; return false; }
        m.dof_act_id[a] = u; m.dof_act_coef[a] = m.act_mval[e]; a++;
      }
    }
    m.dof_act_adr[m.nv] = a;
    for (int e = 0; e < a; e++) m.dof_act_gain[e] = m.act_gain[m.dof_act_id[e]];
    for (int i = 0; i < m.nv; i++) {      // flat per-dof records (dmodel.h: dof_kin, dof_dyn)
      const int j = m.dof_jntid[i], b = m.jnt_bodyid[j];
      m.dof_kin[i][0] = m.jnt_type[j]; m.dof_kin[i][1] = b; m.dof_kin[i][2] = m.body_parentid[b]; m.dof_kin[i][3] = i - m.jnt_dofadr[j];
      float *r = m.dof_dyn[i];
      const int ib[4] = {m.dof_bodyid[i], m.dof_qposadr[i], m.dof_act_adr[i], m.dof_act_adr[i + 1]};
      memcpy(&r[0], &ib[0], 4); r[1] = m.dof_armature[i]; r[2] = m.dof_damping[i]; r[3] = m.dof_stiffness[i];
      memcpy(&r[4], &ib[1], 4); r[5] = m.dof_qspring[i]; memcpy(&r[6], &ib[2], 4); memcpy(&r[7], &ib[3], 4);
      float *q = m.dof_lim[i];
      const int qa = m.jnt_qposadr[j];
      memcpy(&q[0], &qa, 4); q[1] = m.jnt_range[j][0]; q[2] = m.jnt_range[j][1]; q[3] = m.jnt_margin[j];
      q[4] = m.jnt_solref[j][0]; q[5] = m.jnt_solref[j][1];
      for (int k = 0; k < 5; k++) q[6 + k] = m.jnt_solimp[j][k];
      q[11] = m.dof_invweight0[m.jnt_dofadr[j]];
    }
    int g = 0;
    for (int i = 0; i < m.nv; i++) {
      m.dof_grp_adr[i] = g;
      for (int gg = 0; gg < m.ngroup; gg++) for (int d = m.grp_lastdof[gg]; d >= 0; d = m.dof_parentid[d]) if (d == i) m.dof_grp_ids[g++] = gg;
    }
    m.dof_grp_adr[m.nv] = g;
    for (int gg = 0; gg < m.ngroup; gg++) for (int c = m.grp_start[gg]; c < m.grp_start[gg] + m.grp_count[gg]; c++) m.con_grp[c] = gg;
    {  // wrench subsets
      if (m.ncon > 64) { err = "more than 64 contact slots"; return false; }
      std::vector<unsigned long long> masks;
      m.n_wsub = 0;
      for (int i = 0; i < m.nv; i++) {
        unsigned long long cm = 0;
        for (int e = m.dof_grp_adr[i]; e < m.dof_grp_adr[i + 1]; e++) { int gg = m.dof_grp_ids[e]; for (int c = m.grp_start[gg]; c < m.grp_start[gg] + m.grp_count[gg]; c++) cm |= 1ull << c; }
        m.dof_wsub[i] = -1;
        if (!cm) continue;
        size_t s = 0;
        while (s < masks.size() && masks[s] != cm) s++;
        if (s == masks.size()) { masks.push_back(cm); m.wsub_cmask[s][0] = (unsigned)cm; m.wsub_cmask[s][1] = (unsigned)(cm >> 32); }
        m.dof_wsub[i] = (int)s;
      }
      m.n_wsub = (int)masks.size();
      if (m.n_wsub * 6 > 128) { err = "more than 21 distinct paw-wrench subsets (wave kernel limitation)"; return false; }
    }
  }
  for (int i = 0; i < m.nv; i++) {  // two-segment ancestor structure (chain below a trunk chain)
    int sgm = i;
    while (sgm > 0 && m.dof_parentid[sgm] == sgm - 1) sgm--;
    int jump = m.dof_parentid[sgm];
    m.tdof[2 * i] = (m.dof_Madr[i] + m.dof_depth[i]) | (m.dof_depth[i] << 16);
    m.tdof[2 * i + 1] = sgm | ((jump + 1) << 8) | (m.dof_ndesc[i] << 16);
    int r = i - sgm;
    for (int q = 0; q <= m.dof_depth[i]; q++) {
      int a = q <= r ? i - q : jump + 1 + r - q;
      if (a != (int)m.anc_dof[m.dof_Madr[i] + q]) { err = "kinematic tree is not 'chains hanging off one trunk chain' (wave kernel limitation)"; return false; }
    }
    if (m.dof_depth[i] >= 64) { err = "dof depth exceeds the wavefront width"; return false; }
    // packed word of the lean kernel (dmodel.h: tpack); a model whose fields do not fit keeps 0 here and never takes that kernel (rodent_chains_match)
    const int mend = m.dof_Madr[i] + m.dof_depth[i], ws1 = m.dof_wsub[i] + 1;
    m.tpack[i] = (mend < 2048 && m.dof_depth[i] < 64 && r < 32 && ws1 < 16 && jump + 1 == m.dof_depth[i] - r)
                     ? (mend | (m.dof_depth[i] << 11) | (r << 17) | (ws1 << 22)) : 0;
  }
  for (int g = 0; g < m.ngroup; g++) {
    const int ld = m.grp_lastdof[g];
    int sgm = ld;
    while (sgm > 0 && m.dof_parentid[sgm] == sgm - 1) sgm--;
    m.gpack[g] = ld < 0 ? 0xff : (ld | (m.dof_depth[ld] << 8) | ((ld - sgm) << 16));
  }
  for (int b = 0; b < m.nbody; b++) {
    int nsub = 1;
    for (int cb = b + 1; cb < m.nbody; cb++) { bool desc = false; for (int a = m.body_parentid[cb]; a > 0 || a == b; a = m.body_parentid[a]) { if (a == b) { desc = true; break; } if (a == 0) break; } if (desc) nsub++; else break; }
    m.body_nsub[b] = nsub;
    int total = 1;
    for (int cb = 1; cb < m.nbody; cb++) if (cb != b) for (int a = m.body_parentid[cb];; a = m.body_parentid[a]) { if (a == b) { total++; break; } if (a == 0) break; }
    if (b > 0 && total != nsub) { err = "body numbering is not depth-first"; return false; }
  }
  m.n_fix = 0; m.fix_adr[0] = 0;
  for (int p = m.nbody - 1; p >= 1; p--) {
    int nch = 0, first = m.fix_adr[m.n_fix];
    for (int cb = p + 1; cb < p + m.body_nsub[p]; cb += m.body_nsub[cb]) if (cb != p + 1) m.fix_child[first + nch++] = cb;
    if (!nch) continue;
    int r0 = p;
    while (r0 > 1 && m.body_parentid[r0] == r0 - 1) r0--;
    m.fix_p[m.n_fix] = p; m.fix_r0[m.n_fix] = r0; m.fix_adr[++m.n_fix] = first + nch;
  }
  // LDS map of the wave kernel: one source of truth (wave_layout.h); only the total is kept in the model
  m.lds_floats = make_wave_layout(m).lds_floats;
  return true;
}

struct NamedRows { const char *name; int row0, count; bool in_state; };
inline std::vector<NamedRows> debug_rows(const DModel &m) {
  return {
      {"qpos", m.s_qpos, m.nq, true}, {"qvel", m.s_qvel, m.nv, true}, {"act", m.s_act, m.nu, true},
      {"qacc_warmstart", m.s_warm, m.nv, true}, {"time", m.s_time, 1, true}, {"xpos", m.s_xpos, m.nbody * 3, true},
      {"xmat_torso", m.s_xmat_torso, 9, true}, {"qfrc_actuator", m.s_qfrc_actuator, m.nv, true}, {"steps", m.s_steps, 1, true},
      {"xquat", m.w_xquat, m.nbody * 4, false}, {"cinert", m.w_cinert, m.nbody * 10, false}, {"cdof", m.w_cdof, m.nv * 6, false},
      {"qM", m.w_M, m.nnz, false}, {"qfrc_smooth", m.w_qfrc_smooth, m.nv, false}, {"qacc_smooth", m.w_qacc_smooth, m.nv, false},
      {"qacc", m.w_qacc, m.nv, false}, {"qfrc_constraint", m.w_qfrc_constraint, m.nv, false},
      {"con_dist", m.w_con_dist, m.ncon, false}, {"con_frame", m.w_con_frame, m.ncon * 9, false},
      {"efc_D", m.w_efc_D, m.nefc, false}, {"efc_aref", m.w_efc_aref, m.nefc, false}, {"efc_force", m.w_efc_force, m.nefc, false},
      {"subtree_com", m.w_com, 3, false}, {"solver_stats", m.w_solver_stats, 4, false}, {"efc_in", m.w_efc_in, m.nefc, false},
  };
}

}  // namespace tmjx_host
