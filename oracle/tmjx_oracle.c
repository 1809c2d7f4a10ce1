/* oracle/tmjx_oracle.c — CPU ORACLE, physics part (test infrastructure only; see tmjx_oracle.h).
 *
 * Restates mujoco-mjx 3.3.2 `mjx.step` in its *dense* formulation (the reference forces
 * opt.jacobian = dense: track_mjx/environment/task/single_clip_tracking.py:65-72), one env at
 * a time, straightforward loops.  Function names follow mjx/_src/{smooth,collision_primitive,
 * constraint,solver,forward}.py.  PARITY UNPINNED (third-party code not present here).
 */
#include "tmjx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(ORACLE_COUNT)
/* flop-counting build: the R_* wrappers come from counted_real.hpp */
#elif defined(ORACLE_DOUBLE)
#define R_SQRT sqrt
#define R_SIN sin
#define R_COS cos
#define R_POW pow
#define R_FABS fabs
#define R_FMAX fmax
#define R_FMIN fmin
#else
#define R_SQRT sqrtf
#define R_SIN sinf
#define R_COS cosf
#define R_POW powf
#define R_FABS fabsf
#define R_FMAX fmaxf
#define R_FMIN fminf
#endif
#define MJ_MINVAL ((real)1e-15)
#define MJ_MINIMP ((real)0.0001)
#define MJ_MAXIMP ((real)0.9999)

/* ------------------------------------------------------------------ blob reader */
typedef struct { const char *name; int code, count; const void *data; } BEntry;
static int blob_find(const void *blob, size_t n, const char *name, BEntry *out) {
  const unsigned char *p = (const unsigned char *)blob;
  uint32_t magic, ver, ne;
  memcpy(&magic, p, 4); memcpy(&ver, p + 4, 4); memcpy(&ne, p + 8, 4);
  if (magic != 0x584A4D54u || ver != 1) return -1;
  size_t off = 16;
  for (uint32_t i = 0; i < ne && off + 40 <= n; i++) {
    const char *nm = (const char *)(p + off);
    int32_t code, count;
    memcpy(&code, p + off + 32, 4); memcpy(&count, p + off + 36, 4);
    size_t nb = (size_t)count * (code == 0 ? 4 : 8);
    size_t padded = nb + ((8 - nb % 8) % 8);
    if (strncmp(nm, name, 32) == 0) { out->name = nm; out->code = code; out->count = count; out->data = p + off + 40; return 0; }
    off += 40 + padded;
  }
  return -1;
}
static int blob_i(const void *b, size_t n, const char *name, int *dst, int cap) {
  BEntry e; if (blob_find(b, n, name, &e) || e.code != 0) { fprintf(stderr, "oracle: blob entry %s missing\n", name); return -1; }
  if (e.count > cap) { fprintf(stderr, "oracle: blob entry %s too large\n", name); return -1; }
  for (int i = 0; i < e.count; i++) { int32_t v; memcpy(&v, (const char *)e.data + 4 * i, 4); dst[i] = v; }
  return e.count;
}
static int blob_f(const void *b, size_t n, const char *name, real *dst, int cap) {
  BEntry e; if (blob_find(b, n, name, &e) || e.code != 1) { fprintf(stderr, "oracle: blob entry %s missing\n", name); return -1; }
  if (e.count > cap) { fprintf(stderr, "oracle: blob entry %s too large\n", name); return -1; }
  for (int i = 0; i < e.count; i++) { double v; memcpy(&v, (const char *)e.data + 8 * i, 8); dst[i] = (real)v; }
  return e.count;
}
static int blob_f2(const void *b, size_t n, const char *name, real *dst, int rows, int cols, int stride) {
  real *tmp = (real *)malloc(sizeof(real) * (size_t)rows * cols);
  int c = blob_f(b, n, name, tmp, rows * cols);
  if (c != rows * cols) { free(tmp); fprintf(stderr, "oracle: blob entry %s has %d values, expected %d\n", name, c, rows * cols); return -1; }
  for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) dst[i * stride + j] = tmp[i * cols + j];
  free(tmp);
  return 0;
}

OModel *oracle_model_create(const void *blob, size_t n) {
  OModel *m = (OModel *)calloc(1, sizeof(OModel));
  int dims[6];
  if (blob_i(blob, n, "dims", dims, 6) != 6) { free(m); return NULL; }
  m->nbody = dims[0]; m->njnt = dims[1]; m->nq = dims[2]; m->nv = dims[3]; m->nu = dims[4]; m->ncon = dims[5];
  if (m->nbody > O_MAXB || m->nv > O_MAXV || m->nq > O_MAXQ || m->nu > O_MAXU || m->ncon > O_MAXC) { free(m); return NULL; }
  int ok = 1;
#define BI(name, dst, cnt) ok &= (blob_i(blob, n, name, dst, cnt) == (cnt))
#define BF(name, dst, cnt) ok &= (blob_f(blob, n, name, dst, cnt) == (cnt))
#define BF2(name, dst, r, c, s) ok &= (blob_f2(blob, n, name, &dst[0][0], r, c, s) == 0)
  BI("body_parentid", m->body_parentid, m->nbody); BI("body_rootid", m->body_rootid, m->nbody);
  BI("body_jntadr", m->body_jntadr, m->nbody); BI("body_jntnum", m->body_jntnum, m->nbody);
  BI("body_dofadr", m->body_dofadr, m->nbody); BI("body_dofnum", m->body_dofnum, m->nbody);
  BI("jnt_type", m->jnt_type, m->njnt); BI("jnt_bodyid", m->jnt_bodyid, m->njnt);
  BI("jnt_qposadr", m->jnt_qposadr, m->njnt); BI("jnt_dofadr", m->jnt_dofadr, m->njnt);
  BI("jnt_limited", m->jnt_limited, m->njnt);
  BI("dof_bodyid", m->dof_bodyid, m->nv); BI("dof_jntid", m->dof_jntid, m->nv); BI("dof_parentid", m->dof_parentid, m->nv);
  BF2("body_pos", m->body_pos, m->nbody, 3, 3); BF2("body_quat", m->body_quat, m->nbody, 4, 4);
  BF("body_mass", m->body_mass, m->nbody); BF2("body_ipos", m->body_ipos, m->nbody, 3, 3);
  BF2("body_iquat", m->body_iquat, m->nbody, 4, 4); BF2("body_inertia", m->body_inertia, m->nbody, 3, 3);
  BF2("body_invweight0", m->body_invweight0, m->nbody, 2, 2);
  BF2("jnt_pos", m->jnt_pos, m->njnt, 3, 3); BF2("jnt_axis", m->jnt_axis, m->njnt, 3, 3);
  BF2("jnt_range", m->jnt_range, m->njnt, 2, 2); BF("jnt_stiffness", m->jnt_stiffness, m->njnt);
  BF2("jnt_solref", m->jnt_solref, m->njnt, 2, 2); BF2("jnt_solimp", m->jnt_solimp, m->njnt, 5, 5);
  BF("jnt_margin", m->jnt_margin, m->njnt);
  BF("qpos0", m->qpos0, m->nq); BF("qpos_spring", m->qpos_spring, m->nq);
  BF("dof_damping", m->dof_damping, m->nv); BF("dof_armature", m->dof_armature, m->nv);
  BF("dof_invweight0", m->dof_invweight0, m->nv);
  BF2("act_moment", m->act_moment, m->nu, m->nv, O_MAXV); BF("act_gain", m->act_gain, m->nu);
  BF("act_tau", m->act_tau, m->nu); BF2("act_ctrlrange", m->act_ctrlrange, m->nu, 2, 2);
  BF("gravity", m->gravity, 3); BF("meaninertia", &m->meaninertia, 1);
  BI("con_type", m->con_type, m->ncon); BI("con_sub", m->con_sub, m->ncon);
  BI("con_body1", m->con_body1, m->ncon); BI("con_body2", m->con_body2, m->ncon);
  BI("con_geom1", m->con_geom1, m->ncon); BI("con_geom2", m->con_geom2, m->ncon);
  BF2("con_friction", m->con_friction, m->ncon, 3, 3); BF2("con_solref", m->con_solref, m->ncon, 2, 2);
  BF2("con_solimp", m->con_solimp, m->ncon, 5, 5);
  BF2("con_g1_pos", m->con_g1_pos, m->ncon, 3, 3); BF2("con_g1_quat", m->con_g1_quat, m->ncon, 4, 4);
  BF2("con_g2_pos", m->con_g2_pos, m->ncon, 3, 3); BF2("con_g2_quat", m->con_g2_quat, m->ncon, 4, 4);
  BF2("con_g2_size", m->con_g2_size, m->ncon, 3, 3);
  real optf[4]; int opti[3];
  BF("opt_f", optf, 4); BI("opt_i", opti, 3);
  m->timestep = optf[0]; m->tolerance = optf[1]; m->ls_tolerance = optf[2]; m->impratio = optf[3];
  m->iterations = opti[0]; m->ls_iterations = opti[1]; m->n_frames = opti[2];
  int envi[7];
  BI("env_i", envi, 7);
  m->mocap_hz = envi[0]; m->clip_length = envi[1]; m->traj_length = envi[2]; m->window = envi[3];
  m->torso_idx = envi[4]; m->episode_length = envi[5]; m->auto_reset = envi[6]; m->action_repeat = 1;
  m->n_joint_idx = blob_i(blob, n, "joint_idxs", m->joint_idxs, O_MAXV);
  m->n_body_idx = blob_i(blob, n, "body_idxs", m->body_idxs, O_MAXB);
  m->n_endeff_idx = blob_i(blob, n, "endeff_idxs", m->endeff_idxs, 16);
  ok &= m->n_joint_idx > 0 && m->n_body_idx > 0 && m->n_endeff_idx > 0;
  BF("reward_f", m->rw, 25);
  if (!ok || m->window > O_MAXW) { free(m); return NULL; }
  m->nlim = 0;
  for (int j = 0; j < m->njnt; j++) if (m->jnt_limited[j] && m->jnt_type[j] == 3) m->lim_jnt[m->nlim++] = j;
  m->nefc = m->nlim + 4 * m->ncon;
  return m;
}
void oracle_model_destroy(OModel *m) {
  if (!m) return;
  free(m->clip_pos); free(m->clip_quat); free(m->clip_joints); free(m->clip_bodypos); free(m->clip_angvel);
  free(m);
}
size_t oracle_sizeof_data(void) { return sizeof(OData); }

/* ------------------------------------------------------------------ small math (mjx/_src/math.py) */
static inline real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real *o, const real *a, const real *b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real norm3(const real *a) { return R_SQRT(dot3(a, a)); }
static void normalize_n(real *x, int n, real *norm_out) {
  real s = 0; for (int i = 0; i < n; i++) s += x[i] * x[i];
  real nn = R_SQRT(s);
  real den = nn + (real)1e-6 * (nn == (real)0 ? (real)1 : (real)0);
  for (int i = 0; i < n; i++) x[i] = x[i] / den;
  if (norm_out) *norm_out = nn;
}
static void quat_rotate(real *o, const real *v, const real *q) { /* math.rotate(vec, quat) */
  real s = q[0]; const real *u = q + 1; real uv = dot3(u, v), uu = dot3(u, u), c[3];
  cross3(c, u, v);
  for (int i = 0; i < 3; i++) o[i] = 2 * (uv * u[i]) + (s * s - uu) * v[i] + 2 * s * c[i];
}
static void quat_mul(real *o, const real *a, const real *b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void quat_to_mat(real *m, const real *q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void axis_angle_to_quat(real *q, const real *axis, real angle) {
  real s = R_SIN(angle * (real)0.5), c = R_COS(angle * (real)0.5);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void mat_vec3(real *o, const real *m, const real *v) { /* o = M v (row-major 3x3) */
  real a = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], b = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
       c = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void matT_vec3(real *o, const real *m, const real *v) { /* o = M^T v */
  real a = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], b = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
       c = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
/* spatial: motion vectors are [ang(3), lin(3)]; cinert = [xx,yy,zz,xy,xz,yz, m*off(3), m] */
static void inert_mul(real *o, const real *I, const real *v) {
  const real *pos = I + 6; real mass = I[9], c1[3], c2[3];
  real a0 = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
  real a1 = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
  real a2 = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
  cross3(c1, pos, v + 3); cross3(c2, pos, v);
  o[0] = a0 + c1[0]; o[1] = a1 + c1[1]; o[2] = a2 + c1[2];
  o[3] = mass * v[3] - c2[0]; o[4] = mass * v[4] - c2[1]; o[5] = mass * v[5] - c2[2];
}
static void motion_cross(real *o, const real *u, const real *v) {
  real a[3], b[3], c[3];
  cross3(a, u, v); cross3(b, u, v + 3); cross3(c, u + 3, v);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void motion_cross_force(real *o, const real *v, const real *f) {
  real a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

/* ------------------------------------------------------------------ smooth.kinematics */
static void kinematics(const OModel *m, OData *d) {
  for (int k = 0; k < 3; k++) d->xpos[0][k] = 0;
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat_to_mat(d->xmat[0], d->xquat[0]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real pos[3], quat[4], tmp[3];
    quat_rotate(tmp, m->body_pos[b], d->xquat[p]);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[p][k] + tmp[k];
    quat_mul(quat, d->xquat[p], m->body_quat[b]);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, qa = m->jnt_qposadr[j];
      if (m->jnt_type[j] == 0) { /* free: also normalises the quaternion stored in qpos */
        for (int k = 0; k < 3; k++) d->xanchor[j][k] = d->qpos[qa + k];
        d->xaxis[j][0] = 0; d->xaxis[j][1] = 0; d->xaxis[j][2] = 1;
        for (int k = 0; k < 3; k++) pos[k] = d->qpos[qa + k];
        for (int k = 0; k < 4; k++) quat[k] = d->qpos[qa + 3 + k];
        normalize_n(quat, 4, NULL);
        for (int k = 0; k < 4; k++) d->qpos[qa + 3 + k] = quat[k];
      } else { /* hinge */
        real qloc[4], q2[4];
        quat_rotate(tmp, m->jnt_pos[j], quat);
        for (int k = 0; k < 3; k++) d->xanchor[j][k] = tmp[k] + pos[k];
        quat_rotate(d->xaxis[j], m->jnt_axis[j], quat);
        axis_angle_to_quat(qloc, m->jnt_axis[j], d->qpos[qa] - m->qpos0[qa]);
        quat_mul(q2, quat, qloc);
        for (int k = 0; k < 4; k++) quat[k] = q2[k];
        quat_rotate(tmp, m->jnt_pos[j], quat);
        for (int k = 0; k < 3; k++) pos[k] = d->xanchor[j][k] - tmp[k];
      }
    }
    for (int k = 0; k < 3; k++) d->xpos[b][k] = pos[k];
    for (int k = 0; k < 4; k++) d->xquat[b][k] = quat[k];
    quat_to_mat(d->xmat[b], quat);
  }
  for (int b = 0; b < m->nbody; b++) {
    real tmp[3], q[4];
    quat_rotate(tmp, m->body_ipos[b], d->xquat[b]);
    for (int k = 0; k < 3; k++) d->xipos[b][k] = d->xpos[b][k] + tmp[k];
    quat_mul(q, d->xquat[b], m->body_iquat[b]);
    quat_to_mat(d->ximat[b], q);
  }
}

/* ------------------------------------------------------------------ smooth.com_pos */
static void com_pos(const OModel *m, OData *d) {
  static __thread real spos[O_MAXB][3], smass[O_MAXB];
  for (int b = 0; b < m->nbody; b++) {
    for (int k = 0; k < 3; k++) spos[b][k] = d->xipos[b][k] * m->body_mass[b];
    smass[b] = m->body_mass[b];
  }
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 3; k++) spos[p][k] += spos[b][k];
    smass[p] += smass[b];
  }
  for (int b = 0; b < m->nbody; b++)
    for (int k = 0; k < 3; k++) d->subtree_com[b][k] = smass[b] < MJ_MINVAL ? d->xipos[b][k] : spos[b][k] / smass[b];
  for (int b = 0; b < m->nbody; b++) {
    const real *rc = d->subtree_com[m->body_rootid[b]];
    real off[3], mass = m->body_mass[b], I[9];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[b][k] - rc[k];
    /* (ximat * inertia) @ ximat.T */
    const real *X = d->ximat[b], *in = m->body_inertia[b];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      real s = 0; for (int k = 0; k < 3; k++) s += X[i * 3 + k] * in[k] * X[j * 3 + k];
      I[i * 3 + j] = s;
    }
    /* h = cross(off, -eye(3)); h @ h.T * mass  ==  mass * (|off|^2 I - off off^T) */
    real oo = dot3(off, off);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[i * 3 + j] += ((i == j ? oo : (real)0) - off[i] * off[j]) * mass;
    real *c = d->cinert[b];
    c[0] = I[0]; c[1] = I[4]; c[2] = I[8]; c[3] = I[1]; c[4] = I[2]; c[5] = I[5];
    c[6] = off[0] * mass; c[7] = off[1] * mass; c[8] = off[2] * mass; c[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const real *rc = d->subtree_com[m->body_rootid[b]];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = rc[k] - d->xanchor[j][k];
    if (m->jnt_type[j] == 0) {
      for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) d->cdof[da + r][k] = (k == 3 + r) ? 1 : 0;
      for (int r = 0; r < 3; r++) { /* rows of xmat.T = body axes */
        real ax[3] = {d->xmat[b][0 * 3 + r], d->xmat[b][1 * 3 + r], d->xmat[b][2 * 3 + r]}, c[3];
        cross3(c, ax, off);
        for (int k = 0; k < 3; k++) { d->cdof[da + 3 + r][k] = ax[k]; d->cdof[da + 3 + r][3 + k] = c[k]; }
      }
    } else {
      real c[3]; cross3(c, d->xaxis[j], off);
      for (int k = 0; k < 3; k++) { d->cdof[da][k] = d->xaxis[j][k]; d->cdof[da][3 + k] = c[k]; }
    }
  }
}

/* ------------------------------------------------------------------ smooth.crb + factor_m (dense) */
static int cholesky(int n, real A[O_MAXV][O_MAXV], real L[O_MAXV][O_MAXV]) {
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) L[i][j] = 0;
  for (int j = 0; j < n; j++) {
    real s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    L[j][j] = R_SQRT(s);
    for (int i = j + 1; i < n; i++) {
      real t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  return 0;
}
static void cho_solve(int n, real L[O_MAXV][O_MAXV], const real *b, real *x) {
  real y[O_MAXV];
  for (int i = 0; i < n; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = n - 1; i >= 0; i--) { real s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}
static void crb_and_factor(const OModel *m, OData *d) {
  int nv = m->nv;
  for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 10; k++) d->crb[b][k] = d->cinert[b][k];
  for (int b = m->nbody - 1; b >= 1; b--) { int p = m->body_parentid[b]; for (int k = 0; k < 10; k++) d->crb[p][k] += d->crb[b][k]; }
  for (int k = 0; k < 10; k++) d->crb[0][k] = 0;
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->qM[i][j] = 0;
  for (int i = 0; i < nv; i++) {
    real buf[6]; inert_mul(buf, d->crb[m->dof_bodyid[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real s = 0; for (int k = 0; k < 6; k++) s += buf[k] * d->cdof[j][k];
      if (i == j) s += m->dof_armature[i];
      d->qM[i][j] = s; d->qM[j][i] = s;
    }
  }
  cholesky(nv, d->qM, d->qLD);
}
static void mul_m(const OModel *m, const OData *d, const real *v, real *out) {
  for (int i = 0; i < m->nv; i++) { real s = 0; for (int j = 0; j < m->nv; j++) s += d->qM[i][j] * v[j]; out[i] = s; }
}

/* ------------------------------------------------------------------ collision_primitive */
static void plane_sphere(const real *n, const real *ppos, const real *spos, real r, real *dist, real *pos) {
  real df[3]; for (int k = 0; k < 3; k++) df[k] = spos[k] - ppos[k];
  *dist = dot3(df, n) - r;
  for (int k = 0; k < 3; k++) pos[k] = spos[k] - n[k] * (r + (real)0.5 * (*dist));
}
static void make_frame(const real *a_in, real *frame) {
  real a[3] = {a_in[0], a_in[1], a_in[2]}, b[3], c[3];
  normalize_n(a, 3, NULL);
  if ((real)-0.5 < a[1] && a[1] < (real)0.5) { b[0] = 0; b[1] = 1; b[2] = 0; } else { b[0] = 0; b[1] = 0; b[2] = 1; }
  real ab = dot3(a, b);
  for (int k = 0; k < 3; k++) b[k] -= a[k] * ab;
  normalize_n(b, 3, NULL);
  cross3(c, a, b);
  for (int k = 0; k < 3; k++) { frame[k] = a[k]; frame[3 + k] = b[k]; frame[6 + k] = c[k]; }
}
static void collision(const OModel *m, OData *d) {
  for (int c = 0; c < m->ncon; c++) {
    int b1 = m->con_body1[c], b2 = m->con_body2[c];
    real ppos[3], pq[4], pmat[9], gpos[3], gq[4], gmat[9], tmp[3];
    quat_rotate(tmp, m->con_g1_pos[c], d->xquat[b1]);
    for (int k = 0; k < 3; k++) ppos[k] = d->xpos[b1][k] + tmp[k];
    quat_mul(pq, d->xquat[b1], m->con_g1_quat[c]); quat_to_mat(pmat, pq);
    quat_rotate(tmp, m->con_g2_pos[c], d->xquat[b2]);
    for (int k = 0; k < 3; k++) gpos[k] = d->xpos[b2][k] + tmp[k];
    quat_mul(gq, d->xquat[b2], m->con_g2_quat[c]); quat_to_mat(gmat, gq);
    real n[3] = {pmat[2], pmat[5], pmat[8]};
    const real *size = m->con_g2_size[c];
    if (m->con_type[c] == 3) { /* plane_capsule: 2 contacts, slot sub selects the end */
      real axis[3] = {gmat[2], gmat[5], gmat[8]}, b[3], bn, na = dot3(n, axis), cr[3];
      for (int k = 0; k < 3; k++) b[k] = axis[k] - n[k] * na;
      normalize_n(b, 3, &bn);
      if (bn < (real)0.5) {
        if ((real)-0.5 < n[1] && n[1] < (real)0.5) { b[0] = 0; b[1] = 1; b[2] = 0; } else { b[0] = 0; b[1] = 0; b[2] = 1; }
      }
      cross3(cr, n, b);
      for (int k = 0; k < 3; k++) { d->con_frame[c][k] = n[k]; d->con_frame[c][3 + k] = b[k]; d->con_frame[c][6 + k] = cr[k]; }
      real sgn = m->con_sub[c] == 0 ? (real)1 : (real)-1, end[3];
      for (int k = 0; k < 3; k++) end[k] = gpos[k] + sgn * (axis[k] * size[1]);
      plane_sphere(n, ppos, end, size[0], &d->con_dist[c], d->con_pos[c]);
    } else if (m->con_type[c] == 4) { /* plane_ellipsoid */
      real loc[3], sup[3], w[3], pos[3], df[3];
      matT_vec3(loc, gmat, n);
      for (int k = 0; k < 3; k++) sup[k] = loc[k] * size[k];
      normalize_n(sup, 3, NULL);
      for (int k = 0; k < 3; k++) sup[k] = -sup[k] * size[k];
      mat_vec3(w, gmat, sup);
      for (int k = 0; k < 3; k++) { pos[k] = gpos[k] + w[k]; df[k] = pos[k] - ppos[k]; }
      real dist = dot3(n, df);
      for (int k = 0; k < 3; k++) d->con_pos[c][k] = pos[k] - n[k] * dist * (real)0.5;
      d->con_dist[c] = dist;
      make_frame(n, d->con_frame[c]);
    } else { /* plane_sphere */
      plane_sphere(n, ppos, gpos, size[0], &d->con_dist[c], d->con_pos[c]);
      make_frame(n, d->con_frame[c]);
    }
  }
}

/* ------------------------------------------------------------------ constraint.make_constraint */
static void kbi(const OModel *m, const real *solref, const real *solimp, real pos, real *k, real *b, real *imp) {
  real timeconst = solref[0], dampratio = solref[1];
  timeconst = R_FMAX(timeconst, 2 * m->timestep); /* refsafe */
  real dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = R_FMIN(R_FMAX(dmin, MJ_MINIMP), MJ_MAXIMP); dmax = R_FMIN(R_FMAX(dmax, MJ_MINIMP), MJ_MAXIMP);
  width = R_FMAX(MJ_MINVAL, width); mid = R_FMIN(R_FMAX(mid, MJ_MINIMP), MJ_MAXIMP); power = R_FMAX((real)1, power);
  real kk = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  real bb = 2 / (dmax * timeconst);
  if (solref[0] <= 0) kk = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) bb = -solref[1] / dmax;
  real imp_x = R_FABS(pos) / width;
  real imp_a = ((real)1 / R_POW(mid, power - 1)) * R_POW(imp_x, power);
  real imp_b = 1 - ((real)1 / R_POW(1 - mid, power - 1)) * R_POW(1 - imp_x, power);
  real imp_y = imp_x < mid ? imp_a : imp_b;
  real im = dmin + imp_y * (dmax - dmin);
  im = R_FMIN(R_FMAX(im, dmin), dmax);
  if (imp_x > (real)1) im = dmax;
  *k = kk; *b = bb; *imp = im;
}
static void efc_row(const OModel *m, OData *d, int r, real invweight, real pos, const real *solref, const real *solimp) {
  real k, b, imp, vel = 0;
  kbi(m, solref, solimp, pos, &k, &b, &imp);
  real R = R_FMAX(invweight * (1 - imp) / imp, MJ_MINVAL);
  for (int i = 0; i < m->nv; i++) vel += d->efc_J[r][i] * d->qvel[i];
  d->efc_aref[r] = -b * vel - k * imp * pos;
  d->efc_D[r] = 1 / R;
  d->efc_pos[r] = pos;
}
static void make_constraint(const OModel *m, OData *d) {
  int nv = m->nv, r = 0;
  for (int l = 0; l < m->nlim; l++, r++) {
    int j = m->lim_jnt[l], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    real q = d->qpos[qa], dmin = q - m->jnt_range[j][0], dmax = m->jnt_range[j][1] - q;
    real pos = R_FMIN(dmin, dmax) - m->jnt_margin[j];
    int active = pos < 0;
    for (int i = 0; i < nv; i++) d->efc_J[r][i] = 0;
    d->efc_J[r][da] = (real)((dmin < dmax) * 2 - 1) * (real)active;
    efc_row(m, d, r, m->dof_invweight0[da], pos, m->jnt_solref[j], m->jnt_solimp[j]);
  }
  for (int c = 0; c < m->ncon; c++) {
    int b2 = m->con_body2[c], b1 = m->con_body1[c];
    real jacp[3][O_MAXV], off[3], dc[3][O_MAXV];
    for (int k = 0; k < 3; k++) for (int i = 0; i < nv; i++) jacp[k][i] = 0;
    /* support.jac for body2 minus body1; body1 (floor) is static => zero */
    for (int k = 0; k < 3; k++) off[k] = d->con_pos[c][k] - d->subtree_com[m->body_rootid[b2]][k];
    int last = -1;
    for (int b = b2; b > 0 && last < 0; b = m->body_parentid[b]) if (m->body_dofnum[b]) last = m->body_dofadr[b] + m->body_dofnum[b] - 1;
    for (int i = last; i >= 0; i = m->dof_parentid[i]) {
      real cr[3]; cross3(cr, d->cdof[i], off);
      for (int k = 0; k < 3; k++) jacp[k][i] = d->cdof[i][3 + k] + cr[k];
    }
    if (m->body_dofnum[b1] || m->body_parentid[b1] != 0) { fprintf(stderr, "oracle: contact body1 must be static\n"); }
    for (int k = 0; k < 3; k++) for (int i = 0; i < nv; i++)
      dc[k][i] = d->con_frame[c][k * 3 + 0] * jacp[0][i] + d->con_frame[c][k * 3 + 1] * jacp[1][i] + d->con_frame[c][k * 3 + 2] * jacp[2][i];
    real t = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    real pos = d->con_dist[c]; /* includemargin = margin - gap = 0 */
    int active = pos < 0;
    for (int tdir = 0; tdir < 2; tdir++) {
      real fr = m->con_friction[c][0]; /* condim 3: friction[0] serves both tangent directions */
      for (int sg = 0; sg < 2; sg++, r++) {
        real f = sg == 0 ? fr : -fr;
        for (int i = 0; i < nv; i++) d->efc_J[r][i] = (dc[0][i] + dc[1 + tdir][i] * f) * (real)active;
        real invw = (t + f * f * t) * 2 * f * f / m->impratio;
        efc_row(m, d, r, invw, pos, m->con_solref[c], m->con_solimp[c]);
      }
    }
  }
}

/* ------------------------------------------------------------------ fwd_velocity: com_vel, passive, rne */
static void com_vel(const OModel *m, OData *d) {
  for (int k = 0; k < 6; k++) d->cvel[0][k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    real cvel[6]; int p = m->body_parentid[b];
    for (int k = 0; k < 6; k++) cvel[k] = d->cvel[p][k];
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == 0) {
        for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + r][k] * d->qvel[da + r];
        for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) d->cdof_dot[da + r][k] = 0;
        for (int r = 3; r < 6; r++) motion_cross(d->cdof_dot[da + r], cvel, d->cdof[da + r]);
        for (int r = 3; r < 6; r++) for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + r][k] * d->qvel[da + r];
      } else {
        motion_cross(d->cdof_dot[da], cvel, d->cdof[da]);
        for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da][k] * d->qvel[da];
      }
    }
    for (int k = 0; k < 6; k++) d->cvel[b][k] = cvel[k];
  }
}
static void passive(const OModel *m, OData *d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] = 0;
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] != 3) continue;
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    d->qfrc_passive[da] = -m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] += -m->dof_damping[i] * d->qvel[i];
}
static void rne(const OModel *m, OData *d) {
  static __thread real cacc[O_MAXB][6], cfrc[O_MAXB][6];
  for (int k = 0; k < 3; k++) { cacc[0][k] = 0; cacc[0][3 + k] = -m->gravity[k]; }
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 6; k++) cacc[b][k] = cacc[p][k];
    for (int i = 0; i < m->body_dofnum[b]; i++) { int dd = m->body_dofadr[b] + i; for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[dd][k] * d->qvel[dd]; }
  }
  for (int b = 0; b < m->nbody; b++) {
    real f1[6], t[6], f2[6];
    inert_mul(f1, d->cinert[b], cacc[b]);
    inert_mul(t, d->cinert[b], d->cvel[b]);
    motion_cross_force(f2, d->cvel[b], t);
    for (int k = 0; k < 6; k++) cfrc[b][k] = f1[k] + f2[k];
  }
  for (int b = m->nbody - 1; b >= 1; b--) { int p = m->body_parentid[b]; for (int k = 0; k < 6; k++) cfrc[p][k] += cfrc[b][k]; }
  for (int i = 0; i < m->nv; i++) { real s = 0; for (int k = 0; k < 6; k++) s += d->cdof[i][k] * cfrc[m->dof_bodyid[i]][k]; d->qfrc_bias[i] = s; }
}

/* ------------------------------------------------------------------ fwd_actuation / fwd_acceleration */
static void fwd_actuation(const OModel *m, OData *d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_actuator[i] = 0;
  for (int a = 0; a < m->nu; a++) {
    real ctrl = R_FMIN(R_FMAX(d->ctrl[a], m->act_ctrlrange[a][0]), m->act_ctrlrange[a][1]);
    d->act_dot[a] = (ctrl - d->act[a]) / R_FMAX(MJ_MINVAL, m->act_tau[a]);
    real force = m->act_gain[a] * d->act[a];
    d->actuator_force[a] = force;
  }
  for (int i = 0; i < m->nv; i++) { real s = 0; for (int a = 0; a < m->nu; a++) s += m->act_moment[a][i] * d->actuator_force[a]; d->qfrc_actuator[i] = s; }
}
static void fwd_acceleration(const OModel *m, OData *d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  cho_solve(m->nv, d->qLD, d->qfrc_smooth, d->qacc_smooth);
}

/* ------------------------------------------------------------------ solver.solve (CG) */
typedef struct {
  real qacc[O_MAXV], qfrc_constraint[O_MAXV], Jaref[O_MAXEFC], efc_force[O_MAXEFC], Ma[O_MAXV];
  real grad[O_MAXV], Mgrad[O_MAXV], search[O_MAXV], gauss, cost, prev_cost;
  int niter;
} Ctx;
static void update_constraint(const OModel *m, const OData *d, Ctx *c) {
  int nv = m->nv, ne = m->nefc;
  real cost = 0;
  for (int r = 0; r < ne; r++) {
    int active = c->Jaref[r] < 0;
    c->efc_force[r] = d->efc_D[r] * -c->Jaref[r] * (real)active;
    cost += d->efc_D[r] * c->Jaref[r] * c->Jaref[r] * (real)active;
  }
  for (int i = 0; i < nv; i++) { real s = 0; for (int r = 0; r < ne; r++) s += d->efc_J[r][i] * c->efc_force[r]; c->qfrc_constraint[i] = s; }
  real g = 0;
  for (int i = 0; i < nv; i++) g += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
  c->gauss = (real)0.5 * g;
  c->prev_cost = c->cost;
  c->cost = (real)0.5 * cost + c->gauss;
}
static void update_gradient(const OModel *m, const OData *d, Ctx *c) {
  for (int i = 0; i < m->nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  cho_solve(m->nv, (real(*)[O_MAXV])d->qLD, c->grad, c->Mgrad);
}
static void ctx_create(const OModel *m, const OData *d, const real *qacc, Ctx *c, int grad) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) c->qacc[i] = qacc[i];
  for (int r = 0; r < m->nefc; r++) { real s = 0; for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * qacc[i]; c->Jaref[r] = s - d->efc_aref[r]; }
  mul_m(m, d, qacc, c->Ma);
  for (int i = 0; i < nv; i++) { c->grad[i] = 0; c->Mgrad[i] = 0; c->search[i] = 0; }
  c->gauss = 0; c->cost = INFINITY; c->prev_cost = 0; c->niter = 0;
  update_constraint(m, d, c);
  if (grad) { update_gradient(m, d, c); for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i]; }
}
typedef struct { real alpha, cost, deriv_0, deriv_1; } LSPoint;
static LSPoint ls_point(const OModel *m, const Ctx *c, real alpha, const real *jv, const real (*quad)[3], const real *quad_gauss) {
  real qt[3] = {quad_gauss[0], quad_gauss[1], quad_gauss[2]};
  for (int r = 0; r < m->nefc; r++) {
    real x = c->Jaref[r] + alpha * jv[r];
    if (x < 0) { qt[0] += quad[r][0]; qt[1] += quad[r][1]; qt[2] += quad[r][2]; }
  }
  LSPoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * qt[2] + alpha * qt[1] + qt[0];
  p.deriv_0 = 2 * alpha * qt[2] + qt[1];
  p.deriv_1 = 2 * qt[2] + (qt[2] == 0 ? MJ_MINVAL : (real)0);
  return p;
}
static void linesearch(const OModel *m, OData *d, Ctx *c) {
  int nv = m->nv, ne = m->nefc;
  static __thread real mv[O_MAXV], jv[O_MAXEFC], quad[O_MAXEFC][3];
  real sn = 0; for (int i = 0; i < nv; i++) sn += c->search[i] * c->search[i];
  real smag = R_SQRT(sn) * m->meaninertia * (real)(nv > 1 ? nv : 1);
  real gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d, c->search, mv);
  for (int r = 0; r < ne; r++) { real s = 0; for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * c->search[i]; jv[r] = s; }
  real sMa = 0, sq = 0, sMv = 0;
  for (int i = 0; i < nv; i++) { sMa += c->search[i] * c->Ma[i]; sq += c->search[i] * d->qfrc_smooth[i]; sMv += c->search[i] * mv[i]; }
  real quad_gauss[3] = {c->gauss, sMa - sq, (real)0.5 * sMv};
  for (int r = 0; r < ne; r++) {
    quad[r][0] = (real)0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r];
    quad[r][1] = jv[r] * c->Jaref[r] * d->efc_D[r];
    quad[r][2] = (real)0.5 * jv[r] * jv[r] * d->efc_D[r];
  }
  LSPoint p0 = ls_point(m, c, 0, jv, quad, quad_gauss);
  LSPoint lo0 = ls_point(m, c, p0.alpha - p0.deriv_0 / p0.deriv_1, jv, quad, quad_gauss);
  int lesser = lo0.deriv_0 < p0.deriv_0;
  LSPoint hi = lesser ? p0 : lo0, lo = lesser ? lo0 : p0;
  int swap = 1, it = 0;
  for (;;) {
    int done = it >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv_0 < 0) && (lo.deriv_0 > -gtol);
    done |= (hi.deriv_0 > 0) && (hi.deriv_0 < gtol);
    if (done) break;
    LSPoint lo_next = ls_point(m, c, lo.alpha - lo.deriv_0 / lo.deriv_1, jv, quad, quad_gauss);
    LSPoint hi_next = ls_point(m, c, hi.alpha - hi.deriv_0 / hi.deriv_1, jv, quad, quad_gauss);
    LSPoint mid = ls_point(m, c, (real)0.5 * (lo.alpha + hi.alpha), jv, quad, quad_gauss);
    int swap_lo_next = (lo.deriv_0 > 0) || (lo.deriv_0 < lo_next.deriv_0);
    if (swap_lo_next) lo = lo_next;
    int swap_lo_mid = (mid.deriv_0 < 0) && (lo.deriv_0 < mid.deriv_0);
    if (swap_lo_mid) lo = mid;
    int swap_hi_next = (hi.deriv_0 < 0) || (hi.deriv_0 > hi_next.deriv_0);
    if (swap_hi_next) hi = hi_next;
    int swap_hi_mid = (mid.deriv_0 > 0) && (hi.deriv_0 > mid.deriv_0);
    if (swap_hi_mid) hi = mid;
    swap = swap_lo_next | swap_lo_mid | swap_hi_next | swap_hi_mid;
    it++;
  }
  d->ls_total += it;
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  real alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  real ia = (real)improved;
  for (int i = 0; i < nv; i++) { c->qacc[i] = c->qacc[i] + ia * c->search[i] * alpha; c->Ma[i] = c->Ma[i] + ia * mv[i] * alpha; }
  for (int r = 0; r < ne; r++) c->Jaref[r] = c->Jaref[r] + ia * jv[r] * alpha;
}
static void solve(const OModel *m, OData *d) {
  int nv = m->nv;
  static __thread Ctx cw, cs, ctx;
  ctx_create(m, d, d->qacc_warmstart, &cw, 0);
  ctx_create(m, d, d->qacc_smooth, &cs, 0);
  const real *q0 = cw.cost < cs.cost ? d->qacc_warmstart : d->qacc_smooth;
  ctx_create(m, d, q0, &ctx, 1);
  real scale = m->meaninertia * (real)(nv > 1 ? nv : 1);
  for (;;) {
    if (m->iterations != 1) {
      real gn = 0; for (int i = 0; i < nv; i++) gn += ctx.grad[i] * ctx.grad[i];
      real improvement = (ctx.prev_cost - ctx.cost) / scale, gradient = R_SQRT(gn) / scale;
      int done = ctx.niter >= m->iterations;
      done |= improvement < m->tolerance;
      done |= gradient < m->tolerance;
      if (done) break;
    }
    linesearch(m, d, &ctx);
    real pg[O_MAXV], pM[O_MAXV];
    for (int i = 0; i < nv; i++) { pg[i] = ctx.grad[i]; pM[i] = ctx.Mgrad[i]; }
    update_constraint(m, d, &ctx);
    update_gradient(m, d, &ctx);
    real num = 0, den = 0;
    for (int i = 0; i < nv; i++) { num += ctx.grad[i] * (ctx.Mgrad[i] - pM[i]); den += pg[i] * pM[i]; }
    real beta = num / R_FMAX(MJ_MINVAL, den);
    beta = R_FMAX((real)0, beta);
    for (int i = 0; i < nv; i++) ctx.search[i] = -ctx.Mgrad[i] + beta * ctx.search[i];
    ctx.niter++;
    if (m->iterations == 1) break;
  }
  for (int i = 0; i < nv; i++) { d->qacc[i] = ctx.qacc[i]; d->qacc_warmstart[i] = ctx.qacc[i]; d->qfrc_constraint[i] = ctx.qfrc_constraint[i]; }
  for (int r = 0; r < m->nefc; r++) d->efc_force[r] = ctx.efc_force[r];
  d->solver_niter = ctx.niter;
}

/* ------------------------------------------------------------------ forward / euler / step */
void oracle_forward(const OModel *m, OData *d) {
  d->ls_total = 0;
  kinematics(m, d);
  com_pos(m, d);
  crb_and_factor(m, d);
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  passive(m, d);
  rne(m, d);
  fwd_actuation(m, d);
  fwd_acceleration(m, d);
  solve(m, d);
}
static void euler(const OModel *m, OData *d) {
  int nv = m->nv;
  static __thread real Mh[O_MAXV][O_MAXV];
  real qfrc[O_MAXV], qacc[O_MAXV];
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Mh[i][j] = d->qM[i][j];
  for (int i = 0; i < nv; i++) Mh[i][i] += m->timestep * m->dof_damping[i];
  cholesky(nv, Mh, d->qLDh);
  for (int i = 0; i < nv; i++) qfrc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  cho_solve(nv, d->qLDh, qfrc, qacc);
  real h = m->timestep;
  for (int a = 0; a < m->nu; a++) d->act[a] = d->act[a] + d->act_dot[a] * h;
  for (int i = 0; i < nv; i++) d->qvel[i] = d->qvel[i] + qacc[i] * h;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == 0) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] = d->qpos[qa + k] + h * d->qvel[da + k];
      real v[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}, nn, qr[4], q2[4];
      normalize_n(v, 3, &nn);
      axis_angle_to_quat(qr, v, h * nn);
      quat_mul(q2, d->qpos + qa + 3, qr);
      normalize_n(q2, 4, NULL);
      for (int k = 0; k < 4; k++) d->qpos[qa + 3 + k] = q2[k];
    } else {
      d->qpos[qa] = d->qpos[qa] + h * d->qvel[da];
    }
  }
  d->time = d->time + h;
}
void oracle_step(const OModel *m, OData *d) { oracle_forward(m, d); euler(m, d); }

void oracle_data_init(const OModel *m, OData *d, const double *qpos, const double *qvel) {
  memset(d, 0, sizeof(OData));
  for (int i = 0; i < m->nq; i++) d->qpos[i] = (real)qpos[i];
  for (int i = 0; i < m->nv; i++) d->qvel[i] = (real)qvel[i];
}
void oracle_set_ctrl(const OModel *m, OData *d, const double *ctrl) { for (int a = 0; a < m->nu; a++) d->ctrl[a] = (real)ctrl[a]; }

/* named field access for the tests (values widen to double) */
#define FIELD(nm, ptr, cnt) if (!strcmp(name, nm)) { p = (real *)(ptr); n = (cnt); }
static int field_lookup(const OModel *m, OData *d, const char *name, real **pp, int *stride, int *rows, int *cols) {
  real *p = NULL; int n = 0; *stride = 0; *rows = 1;
  FIELD("qpos", d->qpos, m->nq) FIELD("qvel", d->qvel, m->nv) FIELD("act", d->act, m->nu) FIELD("ctrl", d->ctrl, m->nu)
  FIELD("qacc_warmstart", d->qacc_warmstart, m->nv) FIELD("time", &d->time, 1)
  FIELD("xpos", d->xpos, m->nbody * 3) FIELD("xquat", d->xquat, m->nbody * 4) FIELD("xmat", d->xmat, m->nbody * 9)
  FIELD("xipos", d->xipos, m->nbody * 3) FIELD("ximat", d->ximat, m->nbody * 9)
  FIELD("subtree_com", d->subtree_com, m->nbody * 3) FIELD("cinert", d->cinert, m->nbody * 10)
  FIELD("cdof", d->cdof, m->nv * 6) FIELD("cvel", d->cvel, m->nbody * 6) FIELD("cdof_dot", d->cdof_dot, m->nv * 6)
  FIELD("qfrc_bias", d->qfrc_bias, m->nv) FIELD("qfrc_passive", d->qfrc_passive, m->nv)
  FIELD("qfrc_actuator", d->qfrc_actuator, m->nv) FIELD("qfrc_smooth", d->qfrc_smooth, m->nv)
  FIELD("qacc_smooth", d->qacc_smooth, m->nv) FIELD("qacc", d->qacc, m->nv) FIELD("qfrc_constraint", d->qfrc_constraint, m->nv)
  FIELD("act_dot", d->act_dot, m->nu) FIELD("actuator_force", d->actuator_force, m->nu)
  FIELD("con_dist", d->con_dist, m->ncon) FIELD("con_pos", d->con_pos, m->ncon * 3) FIELD("con_frame", d->con_frame, m->ncon * 9)
  FIELD("efc_D", d->efc_D, m->nefc) FIELD("efc_aref", d->efc_aref, m->nefc) FIELD("efc_pos", d->efc_pos, m->nefc)
  FIELD("efc_force", d->efc_force, m->nefc)
  if (!strcmp(name, "qM")) { p = &d->qM[0][0]; *rows = m->nv; *cols = m->nv; *stride = O_MAXV; *pp = p; return m->nv * m->nv; }
  if (!strcmp(name, "efc_J")) { p = &d->efc_J[0][0]; *rows = m->nefc; *cols = m->nv; *stride = O_MAXV; *pp = p; return m->nefc * m->nv; }
  *pp = p; *cols = n;
  return p ? n : -1;
}
int oracle_data_get(const OModel *m, const OData *d, const char *name, double *out, int cap) {
  real *p; int stride, rows, cols;
  if (!strcmp(name, "solver_niter")) { out[0] = d->solver_niter; return 1; }
  if (!strcmp(name, "ls_total")) { out[0] = d->ls_total; return 1; }
  int n = field_lookup(m, (OData *)d, name, &p, &stride, &rows, &cols);
  if (n < 0 || n > cap) return -1;
  if (stride) { for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) out[i * cols + j] = p[i * stride + j]; }
  else for (int i = 0; i < n; i++) out[i] = p[i];
  return n;
}
int oracle_data_set(const OModel *m, OData *d, const char *name, const double *in, int n_in) {
  real *p; int stride, rows, cols;
  int n = field_lookup(m, d, name, &p, &stride, &rows, &cols);
  if (n < 0 || n != n_in || stride) return -1;
  for (int i = 0; i < n; i++) p[i] = (real)in[i];
  return n;
}
