/* oracle/tmjx_oracle_abi.c — the CPU oracle behind the SAME C-ABI as libtmjx_hip.so (include/tmjx.h), on HOST buffers.
 *
 * TEST INFRASTRUCTURE ("libtrackmjx_cpu.so = oracle with identical symbols"): one harness — the ctypes
 * binding of track_mjx_amd/hip.py, `hip.load(path)` — drives the HIP library and this one with the same calls, the same
 * [row][n_env] structure-of-arrays buffers and the same row layout (tmjx_layout), so outputs can be diffed buffer against buffer.
 * Never loaded by the product.  Entry points: model create / destroy / layout / set_wrappers, clips_upload, reset, step,
 * physics, forward, reward_obs, gae, last_error, version; the learner kernels (GEMMs, loss head, optimiser ...) have no oracle
 * twin here (their checker is float64 torch in tests/).  `stream` is ignored; every call is synchronous, one env after the other.
 */
#include "../include/tmjx.h"
#include "tmjx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct tmjx_model { OModel *o; tmjx_layout_t L; };
static __thread char g_err[256];
static int fail(int code, const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
const char *tmjx_last_error(void) { return g_err; }
const char *tmjx_version(void) { return "tmjx-oracle-abi 0.1 (CPU restatement, test infrastructure)"; }

static void make_layout(const OModel *m, tmjx_layout_t *L) {
  memset(L, 0, sizeof *L);
  L->nq = m->nq; L->nv = m->nv; L->nu = m->nu; L->nbody = m->nbody; L->ncon = m->ncon; L->nefc = m->nefc;
  L->obs_size = oracle_obs_size(m);
  int T = m->traj_length;
  L->ref_obs_size = T * 3 + T * 4 + T * m->n_joint_idx + T * m->n_body_idx * 3;
  L->n_metrics = 20; L->window = m->window;
  int s = 0;                                    /* the row order of csrc/model_host.h */
  L->qpos = s; s += m->nq; L->qvel = s; s += m->nv; L->act = s; s += m->nu; L->qacc_warmstart = s; s += m->nv; L->time = s; s += 1;
  int nphys = s;
  L->xpos = s; s += m->nbody * 3; L->xmat_torso = s; s += 9; L->qfrc_actuator = s; s += m->nv;
  L->prev_ctrl = s; s += m->nu; L->action_buffer = s; s += m->window * m->nu; L->done = s; s += 1; L->steps_f = s; s += 1;
  L->first_phys = s; s += nphys; L->first_obs = s; s += L->obs_size; L->first_prev_ctrl = s; s += m->nu;
  L->state_rows = s;
  L->i_clip_idx = 0; L->i_start_frame = 1; L->i_buffer_index = 2; L->i_nan_count = 3; L->istate_rows = 4;
  L->ws_rows = 1;
}

int tmjx_model_create(const void *blob, size_t nbytes, tmjx_model **out) {
  if (!blob || !out) return fail(TMJX_EINVAL, "null argument");
  OModel *o = oracle_model_create(blob, nbytes);
  if (!o) return fail(TMJX_EINVAL, "oracle_model_create failed");
  tmjx_model *m = (tmjx_model *)calloc(1, sizeof *m);
  m->o = o;
  make_layout(o, &m->L);
  *out = m;
  return TMJX_OK;
}
void tmjx_model_destroy(tmjx_model *m) { if (m) { oracle_model_destroy(m->o); free(m); } }
int tmjx_layout(const tmjx_model *m, tmjx_layout_t *out) { if (!m || !out) return fail(TMJX_EINVAL, "null argument"); *out = m->L; return TMJX_OK; }
int tmjx_set_wrappers(tmjx_model *m, int episode_length, int auto_reset) {
  if (!m || episode_length < 1) return fail(TMJX_EINVAL, "bad argument");
  m->o->episode_length = episode_length; m->o->auto_reset = auto_reset ? 1 : 0;
  return TMJX_OK;
}
int tmjx_set_action_repeat(tmjx_model *m, int action_repeat) {
  if (!m || action_repeat < 1) return fail(TMJX_EINVAL, "bad argument");
  oracle_set_action_repeat(m->o, action_repeat);
  return TMJX_OK;
}
int tmjx_clips_upload(tmjx_model *m, const float *position, const float *quaternion, const float *joints, const float *body_positions,
                      const float *angular_velocity, int n_clips, int n_frames) {
  if (!m || !position || !quaternion || !joints || !body_positions || !angular_velocity) return fail(TMJX_EINVAL, "null argument");
  return oracle_set_clips(m->o, position, quaternion, joints, body_positions, angular_velocity, n_clips, n_frames) ? fail(TMJX_EINVAL, "oracle_set_clips") : TMJX_OK;
}

#define ROW(buf, row) ((buf) + (size_t)(row) * (size_t)n + (size_t)e)
static void data_from_rows(const tmjx_model *mm, OData *d, const float *st, int row0, int n, int e, int with_outputs) {
  const OModel *m = mm->o; const tmjx_layout_t *L = &mm->L;
  int off = row0 - L->qpos;                     /* 0 for the live state, first_phys - qpos for the snapshot */
  for (int i = 0; i < m->nq; i++) d->qpos[i] = (real)*ROW(st, L->qpos + off + i);
  for (int i = 0; i < m->nv; i++) d->qvel[i] = (real)*ROW(st, L->qvel + off + i);
  for (int i = 0; i < m->nu; i++) d->act[i] = (real)*ROW(st, L->act + off + i);
  for (int i = 0; i < m->nv; i++) d->qacc_warmstart[i] = (real)*ROW(st, L->qacc_warmstart + off + i);
  d->time = (real)*ROW(st, L->time + off);
  if (with_outputs) {
    for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 3; k++) d->xpos[b][k] = (real)*ROW(st, L->xpos + b * 3 + k);
    for (int k = 0; k < 9; k++) d->xmat[m->torso_idx][k] = (real)*ROW(st, L->xmat_torso + k);
    for (int i = 0; i < m->nv; i++) d->qfrc_actuator[i] = (real)*ROW(st, L->qfrc_actuator + i);
  }
}
static void data_to_rows(const tmjx_model *mm, const OData *d, float *st, int n, int e) {
  const OModel *m = mm->o; const tmjx_layout_t *L = &mm->L;
  for (int i = 0; i < m->nq; i++) *ROW(st, L->qpos + i) = (float)d->qpos[i];
  for (int i = 0; i < m->nv; i++) *ROW(st, L->qvel + i) = (float)d->qvel[i];
  for (int i = 0; i < m->nu; i++) *ROW(st, L->act + i) = (float)d->act[i];
  for (int i = 0; i < m->nv; i++) *ROW(st, L->qacc_warmstart + i) = (float)d->qacc_warmstart[i];
  *ROW(st, L->time) = (float)d->time;
  for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 3; k++) *ROW(st, L->xpos + b * 3 + k) = (float)d->xpos[b][k];
  for (int k = 0; k < 9; k++) *ROW(st, L->xmat_torso + k) = (float)d->xmat[m->torso_idx][k];
  for (int i = 0; i < m->nv; i++) *ROW(st, L->qfrc_actuator + i) = (float)d->qfrc_actuator[i];
}
static void env_from_rows(const tmjx_model *mm, OEnv *ev, const float *st, const int32_t *is, int n, int e) {
  const OModel *m = mm->o; const tmjx_layout_t *L = &mm->L;
  memset(ev, 0, sizeof *ev);
  data_from_rows(mm, &ev->d, st, L->qpos, n, e, 1);
  data_from_rows(mm, &ev->first_d, st, L->first_phys, n, e, 0);
  ev->clip_idx = *ROW(is, L->i_clip_idx); ev->start_frame = *ROW(is, L->i_start_frame); ev->buffer_index = *ROW(is, L->i_buffer_index);
  for (int i = 0; i < m->nu; i++) { ev->prev_ctrl[i] = (real)*ROW(st, L->prev_ctrl + i); ev->first_prev_ctrl[i] = (real)*ROW(st, L->first_prev_ctrl + i); }
  for (int r = 0; r < m->window; r++) for (int i = 0; i < m->nu; i++) ev->action_buffer[r][i] = (real)*ROW(st, L->action_buffer + r * m->nu + i);
  ev->done = (real)*ROW(st, L->done); ev->steps = (real)*ROW(st, L->steps_f);
  for (int i = 0; i < L->obs_size; i++) ev->first_obs[i] = (real)*ROW(st, L->first_obs + i);
}
static void env_to_rows(const tmjx_model *mm, const OEnv *ev, float *st, int32_t *is, int n, int e, int snapshot) {
  const OModel *m = mm->o; const tmjx_layout_t *L = &mm->L;
  data_to_rows(mm, &ev->d, st, n, e);
  *ROW(is, L->i_clip_idx) = ev->clip_idx; *ROW(is, L->i_start_frame) = ev->start_frame; *ROW(is, L->i_buffer_index) = ev->buffer_index;
  for (int i = 0; i < m->nu; i++) *ROW(st, L->prev_ctrl + i) = (float)ev->prev_ctrl[i];
  for (int r = 0; r < m->window; r++) for (int i = 0; i < m->nu; i++) *ROW(st, L->action_buffer + r * m->nu + i) = (float)ev->action_buffer[r][i];
  *ROW(st, L->done) = (float)ev->done; *ROW(st, L->steps_f) = (float)ev->steps;
  if (snapshot) {
    const OData *f = &ev->first_d;
    int o = L->first_phys;
    for (int i = 0; i < m->nq; i++) *ROW(st, o++) = (float)f->qpos[i];
    for (int i = 0; i < m->nv; i++) *ROW(st, o++) = (float)f->qvel[i];
    for (int i = 0; i < m->nu; i++) *ROW(st, o++) = (float)f->act[i];
    for (int i = 0; i < m->nv; i++) *ROW(st, o++) = (float)f->qacc_warmstart[i];
    *ROW(st, o) = (float)f->time;
    for (int i = 0; i < L->obs_size; i++) *ROW(st, L->first_obs + i) = (float)ev->first_obs[i];
    for (int i = 0; i < m->nu; i++) *ROW(st, L->first_prev_ctrl + i) = (float)ev->first_prev_ctrl[i];
  }
}
static void outputs_to_rows(const tmjx_model *mm, const OEnv *ev, float *obs, float *reward, float *done, float *trunc, float *metrics, int n, int e) {
  if (obs) for (int i = 0; i < mm->L.obs_size; i++) *ROW(obs, i) = (float)ev->obs[i];
  if (reward) reward[e] = (float)ev->reward;
  if (done) done[e] = (float)ev->done;
  if (trunc) trunc[e] = (float)ev->truncation;
  if (metrics) for (int k = 0; k < 20; k++) *ROW(metrics, k) = (float)ev->metrics[k];
}

int tmjx_reset(tmjx_model *m, float *state, int32_t *istate, const int32_t *clip_idx, const int32_t *start_frame, const float *qpos_noise,
               const float *qvel_noise, float *obs, float *workspace, int n, void *stream) {
  (void)workspace; (void)stream;
  if (!m || !state || !istate || !clip_idx || !start_frame || !qpos_noise || !qvel_noise || !obs) return fail(TMJX_EINVAL, "null argument");
  if (n < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  OEnv *ev = (OEnv *)malloc(sizeof(OEnv));
  double qn[O_MAXQ], vn[O_MAXV];
  for (int e = 0; e < n; e++) {
    for (int i = 0; i < m->o->nq; i++) qn[i] = *ROW(qpos_noise, i);
    for (int i = 0; i < m->o->nv; i++) vn[i] = *ROW(qvel_noise, i);
    memset(ev, 0, sizeof *ev);
    oracle_env_reset(m->o, ev, clip_idx[e], start_frame[e], qn, vn);
    env_to_rows(m, ev, state, istate, n, e, 1);
    *ROW(istate, m->L.i_nan_count) = 0;
    outputs_to_rows(m, ev, obs, NULL, NULL, NULL, NULL, n, e);
  }
  free(ev);
  return TMJX_OK;
}
typedef struct { const float *pos, *quat, *joints, *bodypos, *angvel; } FrameLeaves;      /* per env row-major: [n][3 | 4 | nq-7 | (nbody-1)*3 | 3] */
static int step_impl(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done, float *trunc,
                     float *metrics, int n, int do_physics, const FrameLeaves *fo) {
  if (!m || !state || !istate || !action || !obs || !reward || !done || !trunc || !metrics) return fail(TMJX_EINVAL, "null argument");
  if (n < 1) return fail(TMJX_EINVAL, "n_env must be >= 1");
  OEnv *ev = (OEnv *)malloc(sizeof(OEnv));
  double a[O_MAXU];
  for (int e = 0; e < n; e++) {
    env_from_rows(m, ev, state, istate, n, e);
    for (int i = 0; i < m->o->nu; i++) a[i] = *ROW(action, i);
    ev->fo_pos = NULL;
    if (fo) {
      const int nj = m->o->nq - 7, nb3 = (m->o->nbody - 1) * 3;
      ev->fo_pos = fo->pos + (size_t)e * 3; ev->fo_quat = fo->quat + (size_t)e * 4; ev->fo_joints = fo->joints + (size_t)e * nj;
      ev->fo_bodypos = fo->bodypos + (size_t)e * nb3; ev->fo_angvel = fo->angvel + (size_t)e * 3;
    }
    oracle_env_step_ex(m->o, ev, a, do_physics);
    env_to_rows(m, ev, state, istate, n, e, 0);
    outputs_to_rows(m, ev, obs, reward, done, trunc, metrics, n, e);
  }
  free(ev);
  return TMJX_OK;
}
int tmjx_step(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done, float *truncation,
              float *metrics, float *workspace, int n_env, void *stream) {
  (void)workspace; (void)stream;
  return step_impl(m, state, istate, action, obs, reward, done, truncation, metrics, n_env, 1, NULL);
}
int tmjx_reward_obs(tmjx_model *m, float *state, int32_t *istate, const float *action, float *obs, float *reward, float *done, float *truncation,
                    float *metrics, float *workspace, int n_env, void *stream) {
  (void)workspace; (void)stream;
  return step_impl(m, state, istate, action, obs, reward, done, truncation, metrics, n_env, 0, NULL);
}
/* include/tmjx.h: K3's reward / termination part with the CALLER's reference frame per env (reward.py:359-366) */
int tmjx_reward_frame(tmjx_model *m, float *state, int32_t *istate, const float *action, const float *frame_pos, const float *frame_quat,
                      const float *frame_joints, const float *frame_bodypos, const float *frame_angvel, float *obs, float *reward, float *done,
                      float *truncation, float *metrics, int n_env, void *stream) {
  (void)stream;
  if (!frame_pos || !frame_quat || !frame_joints || !frame_bodypos || !frame_angvel) return fail(TMJX_EINVAL, "null reference-frame leaf");
  FrameLeaves fo = {frame_pos, frame_quat, frame_joints, frame_bodypos, frame_angvel};
  return step_impl(m, state, istate, action, obs, reward, done, truncation, metrics, n_env, 0, &fo);
}
int tmjx_physics(tmjx_model *m, float *state, const float *action, int n_substeps, float *workspace, int n, void *stream) {
  (void)workspace; (void)stream;
  if (!m || !state || n < 1 || n_substeps < 0) return fail(TMJX_EINVAL, "bad argument");
  OData *d = (OData *)calloc(1, sizeof(OData));
  double a[O_MAXU];
  for (int e = 0; e < n; e++) {
    memset(d, 0, sizeof *d);
    data_from_rows(m, d, state, m->L.qpos, n, e, 0);
    for (int i = 0; i < m->o->nu; i++) a[i] = action ? *ROW(action, i) : 0.0;
    oracle_set_ctrl(m->o, d, a);
    for (int s = 0; s < n_substeps; s++) oracle_step(m->o, d);
    data_to_rows(m, d, state, n, e);
  }
  free(d);
  return TMJX_OK;
}
int tmjx_physics_step(tmjx_model *m, float *state, const float *action, float *workspace, int n_env, void *stream) {
  return tmjx_physics(m, state, action, m ? m->o->n_frames : 0, workspace, n_env, stream);
}
int tmjx_forward(tmjx_model *m, float *state, float *workspace, int n, void *stream) {
  (void)workspace; (void)stream;
  if (!m || !state || n < 1) return fail(TMJX_EINVAL, "bad argument");
  OData *d = (OData *)calloc(1, sizeof(OData));
  for (int e = 0; e < n; e++) {
    memset(d, 0, sizeof *d);
    data_from_rows(m, d, state, m->L.qpos, n, e, 0);
    oracle_forward(m->o, d);
    data_to_rows(m, d, state, n, e);
  }
  free(d);
  return TMJX_OK;
}
int tmjx_gae(const float *truncation, const float *termination, const float *rewards, const float *values, const float *bootstrap, float lambda_,
             float discount, float *vs, float *advantages, int T, int B, void *stream) {
  (void)stream;
  if (!truncation || !termination || !rewards || !values || !bootstrap || !vs || !advantages || T < 1 || B < 1) return fail(TMJX_EINVAL, "bad argument");
  size_t nb = (size_t)T * B;
  double *buf = (double *)malloc(sizeof(double) * (6 * nb + B));
  double *tr = buf, *te = tr + nb, *rw = te + nb, *va = rw + nb, *o1 = va + nb, *o2 = o1 + nb, *bo = o2 + nb;
  for (size_t i = 0; i < nb; i++) { tr[i] = truncation[i]; te[i] = termination[i]; rw[i] = rewards[i]; va[i] = values[i]; }
  for (int b = 0; b < B; b++) bo[b] = bootstrap[b];
  oracle_gae(tr, te, rw, va, bo, lambda_, discount, o1, o2, T, B);
  for (size_t i = 0; i < nb; i++) { vs[i] = (float)o1[i]; advantages[i] = (float)o2[i]; }
  free(buf);
  return TMJX_OK;
}
