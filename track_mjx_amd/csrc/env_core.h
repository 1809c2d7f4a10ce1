// csrc/env_core.h — per-env task logic of the hot path (K1 reset tail, K3 reward/obs/done/wrappers).
//
// Replaces, per env (paths relative to /root/reference/track_mjx):
//   environment/task/single_clip_tracking.py:220-320  step() after pipeline_step
//   environment/task/single_clip_tracking.py:322-454  _get_proprioception/_get_reference_trajectory/_get_obs/_get_cur_frame
//   environment/task/reward.py:57-485                  the 18 reward / cost / termination terms
//   environment/walker/base.py:170-258                 egocentric observation maths (brax.math.rotate / relative_quat)
//   environment/wrappers.py:104-144                    auto-reset (+ brax EpisodeWrapper steps/truncation)
// Quirks reproduced on purpose: prev_ctrl is overwritten before the reward call (ctrl_diff_cost == 0),
// xpos[1:][body id] off-by-one with index clamp, joint id - 1, action window / clip / start frame survive auto-reset.
#pragma once
#include <float.h>

#include "tm_common.h"

#define IS(off) r_is[(size_t)(off) * (size_t)r.n + (size_t)r.e]
#define OUTROW(buf, row) (buf)[(size_t)(row) * (size_t)r.n + (size_t)r.e]

#ifdef TM_HOST_EMU
TM_DEV float tm_mul_noc(float a, float b) { volatile float t = a * b; return t; }
TM_DEV float tm_add_noc(float a, float b) { volatile float t = a + b; return t; }
#else
TM_DEV float tm_mul_noc(float a, float b) { return __fmul_rn(a, b); }
TM_DEV float tm_add_noc(float a, float b) { return __fadd_rn(a, b); }
#endif

TM_DEV int tm_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
// single_clip_tracking.py:452-454: floor(time * mocap_hz + start_frame) in fp32, multiply and add NOT fused
TM_DEV int tm_cur_frame(const DModel &m, float time, int start) {
  return (int)floorf(tm_add_noc(tm_mul_noc(time, (float)m.mocap_hz), (float)start));
}
TM_DEV size_t tm_clip_row(const DModel &m, int clip, int frame) {
  clip = tm_clampi(clip, 0, m.n_clips - 1);
  frame = tm_clampi(frame, 0, m.n_frames_clip - 1);
  return (size_t)clip * (size_t)m.n_frames_clip + (size_t)frame;
}
TM_DEV float tm_nan_to_num(float x) {
  if (x != x) return 0.f;
  if (x > FLT_MAX) return FLT_MAX;
  if (x < -FLT_MAX) return -FLT_MAX;
  return x;
}

// _get_obs -> obs[obs_size][n]; applies nan_to_num when `sanitize`.  `part` < 0: everything.  Otherwise one of the
// TM_OBS_PARTS(T) = 3 T + 3 pieces the split kernel k_obs runs in parallel (one per blockIdx.y): 3 t + {0: root position +
// quaternion, 1: joints, 2: bodies} of trajectory frame t, then 3 T + {0: qpos, 1: qvel, 2: actuator forces + torso + end effectors}.
#define TM_OBS_PARTS(T) (3 * (T) + 3)
// (`obs` is restrict: the observation rows overlap nothing this function reads — without that every PUT orders the loads behind it, and the
// 67-element proprioception pieces were 67 load -> store round trips in a row)
TM_DEV void tm_get_obs(const DModel &m, EnvRef r, int clip, int frame, float *__restrict__ obs, bool sanitize, int part = -1) {
  int nj = m.nq - 7, nbp = m.nbody - 1, T = m.traj_length, o = 0;
  int start = tm_clampi(frame + 1, 0, m.n_frames_clip - T);
  float root[3], quat[4];
  TM_LD(root, ST, m.s_qpos, 0, 3);
  TM_LD(quat, ST, m.s_qpos, 3, 4);
#define PUT(v) do { float v_ = (v); OUTROW(obs, o) = sanitize ? tm_nan_to_num(v_) : v_; o++; } while (0)
#define WANT(p) (part < 0 || part == (p))
  for (int t = 0; t < T; t++) {
    if (part >= 0 && part / 3 != t) continue;
    if (WANT(3 * t)) {
      o = 3 * t;
      const float *rp = m.clip_pos + tm_clip_row(m, clip, start + t) * 3;
      float v[3] = {rp[0] - root[0], rp[1] - root[1], rp[2] - root[2]}, w[3];
      tm_rotate(w, v, quat);
      PUT(w[0]); PUT(w[1]); PUT(w[2]);
      o = 3 * T + 4 * t;
      const float *rq = m.clip_quat + tm_clip_row(m, clip, start + t) * 4;
      float inv[4] = {rq[0], -rq[1], -rq[2], -rq[3]}, wq[4];
      tm_quat_mul(wq, quat, inv);
      PUT(wq[0]); PUT(wq[1]); PUT(wq[2]); PUT(wq[3]);
    }
    if (WANT(3 * t + 1)) {
      o = 7 * T + m.n_joint_idx * t;
      const float *rj = m.clip_joints + tm_clip_row(m, clip, start + t) * nj;
      for (int k = 0; k < m.n_joint_idx; k++) { int i = tm_clampi(m.joint_idxs[k] - 1, 0, nj - 1); PUT(rj[i] - ST(m.s_qpos, 7 + i)); }
    }
    if (WANT(3 * t + 2)) {
      o = 7 * T + m.n_joint_idx * T + 3 * m.n_body_idx * t;
      const float *rb = m.clip_bodypos + tm_clip_row(m, clip, start + t) * (size_t)(nbp * 3);
      for (int k = 0; k < m.n_body_idx; k++) {
        int i = tm_clampi(m.body_idxs[k], 0, nbp - 1);
        float vb[3] = {rb[i * 3] - ST(m.s_xpos, (1 + i) * 3), rb[i * 3 + 1] - ST(m.s_xpos, (1 + i) * 3 + 1), rb[i * 3 + 2] - ST(m.s_xpos, (1 + i) * 3 + 2)}, wb[3];
        tm_rotate(wb, vb, quat);
        PUT(wb[0]); PUT(wb[1]); PUT(wb[2]);
      }
    }
  }
  if (part >= 0 && part < 3 * T) return;
  const int o0 = T * (7 + m.n_joint_idx + 3 * m.n_body_idx);
  if (WANT(3 * T)) { o = o0; for (int i = 7; i < m.nq; i++) PUT(ST(m.s_qpos, i)); }
  if (WANT(3 * T + 1)) { o = o0 + (m.nq - 7); for (int i = 6; i < m.nv; i++) PUT(ST(m.s_qvel, i)); }
  if (WANT(3 * T + 2)) {
    o = o0 + (m.nq - 7) + (m.nv - 6);
    for (int i = 0; i < m.nv; i++) PUT(ST(m.s_qfrc_actuator, i));
    int tb = m.torso_idx;
    float tp[3], X[9];
    TM_LD(tp, ST, m.s_xpos, tb * 3, 3);
    TM_LD(X, ST, m.s_xmat_torso, 0, 9);
    PUT(tp[2]); PUT(X[6]); PUT(X[7]); PUT(X[8]);
    for (int k = 0; k < m.n_endeff_idx; k++) {
      int b = m.endeff_idxs[k];
      float v[3] = {ST(m.s_xpos, b * 3) - tp[0], ST(m.s_xpos, b * 3 + 1) - tp[1], ST(m.s_xpos, b * 3 + 2) - tp[2]};
      for (int c = 0; c < 3; c++) PUT(v[0] * X[c] + v[1] * X[3 + c] + v[2] * X[6 + c]);
    }
  }
#undef WANT
#undef PUT
}

TM_DEV bool tm_state_has_nan(const DModel &m, EnvRef r) {
  bool bad = false;
  for (int i = 0; i < m.nphys; i++) { float v = ST(m.s_qpos, i); bad |= (v != v); }
  for (int i = 0; i < m.nbody * 3; i++) { float v = ST(m.s_xpos, i); bad |= (v != v); }
  for (int i = 0; i < m.nv; i++) { float v = ST(m.s_qfrc_actuator, i); bad |= (v != v); }
  return bad;
}

// Everything of wrappers.wrap(env).step around/after the physics.  Call tm_step_prologue BEFORE the physics.
TM_DEV void tm_step_prologue(const DModel &m, EnvRef r) {
  if (ST(m.s_done, 0) != 0.f) ST(m.s_steps, 0) = 0.f;
  ST(m.s_done, 0) = 0.f;
}
// Windowed action statistics of ONE action dimension i of one env (reward.py:314-356): writes the action into the ring
// buffer row `bi`, then returns the variance term and the jerk term of column i with the advanced index.
TM_DEV void tm_window_dim(const DModel &m, EnvRef r, int i, int bi, float a, float &var_i, float &jerk_i) {
  int nu = m.nu, W = m.window;
  ST(m.s_action_buffer, bi * nu + i) = a;
  bi = (bi + 1) % W;
  float mean = 0.f, var = 0.f, jerk = 0.f;
  for (int rr = 0; rr < W; rr++) mean += ST(m.s_action_buffer, rr * nu + i);
  mean /= (float)W;
  for (int rr = 0; rr < W; rr++) { float df = ST(m.s_action_buffer, rr * nu + i) - mean; var += df * df; }
  int i0 = bi % W, i1 = (bi + 1) % W;
  float o0 = ST(m.s_action_buffer, i0 * nu + i), o1 = ST(m.s_action_buffer, i1 * nu + i);
  for (int rr = 0; rr + 2 < W; rr++) {
    int i2 = (bi + rr + 2) % W;
    float o2 = ST(m.s_action_buffer, i2 * nu + i), j = o2 - 2.f * o1 + o0;
    jerk += j * j;
    o0 = o1; o1 = o2;
  }
  var_i = var / (float)W; jerk_i = jerk;
}
// The four long sums of tm_step_post as separately launchable parts (k_post_parts, one lane per (env, part)); results go to
// rows 0..TM_NPOST-1 of `P` ([row][n]).  Same expressions and the same summation order as the inline code below.
#define TM_NPOST 7   // joint distance | body-position sum | end-effector sum | energy sum | NaN flags of three state thirds
TM_DEV void tm_post_part(const DModel &m, EnvRef r, const int *r_is, int part, float *P) {
  int nj = m.nq - 7, nbp = m.nbody - 1;
  if (part <= 2) {
    int clip = IS(m.i_clip_idx), start = IS(m.i_start_frame);
    size_t row = tm_clip_row(m, clip, tm_cur_frame(m, ST(m.s_time, 0), start));
    const float *rj = m.clip_joints + row * nj, *rb = m.clip_bodypos + row * (size_t)(nbp * 3);
    float s = 0.f;
    if (part == 0) {
#pragma unroll 8
      for (int k = 0; k < nj; k++) { float df = ST(m.s_qpos, 7 + k) - rj[k]; s += df * df; }
      OUTROW(P, 0) = s;
    } else if (part == 1) {
      for (int k = 0; k < m.n_body_idx; k++) {
        int i = tm_clampi(m.body_idxs[k], 0, nbp - 1);
        for (int c = 0; c < 3; c++) { float df = ST(m.s_xpos, (1 + i) * 3 + c) - rb[i * 3 + c]; s += df * df; }
      }
      OUTROW(P, 1) = s;
    } else {
      for (int k = 0; k < m.n_endeff_idx; k++) {
        int i = tm_clampi(m.endeff_idxs[k], 0, nbp - 1);
        for (int c = 0; c < 3; c++) { float df = ST(m.s_xpos, (1 + i) * 3 + c) - rb[i * 3 + c]; s += df * df; }
      }
      OUTROW(P, 2) = s;
    }
  } else if (part == 3) {
    float s = 0.f;
    for (int i = 6; i < m.nv; i++) s += fabsf(ST(m.s_qvel, i)) * fabsf(ST(m.s_qfrc_actuator, i));
    OUTROW(P, 3) = s;
  } else {
    bool bad = false;
    if (part == 4) {
#pragma unroll 8
      for (int i = 0; i < m.nphys; i++) { float v = ST(m.s_qpos, i); bad |= (v != v); }
    } else if (part == 5) {
#pragma unroll 8
      for (int i = 0; i < m.nbody * 3; i++) { float v = ST(m.s_xpos, i); bad |= (v != v); }
    } else {
#pragma unroll 8
      for (int i = 0; i < m.nv; i++) { float v = ST(m.s_qfrc_actuator, i); bad |= (v != v); }
    }
    OUTROW(P, part) = bad ? 1.f : 0.f;
  }
}
// `rep`: which of brax EpisodeWrapper's `action_repeat` inner steps this call is (wrappers.py:43; EpisodeWrapper.step [3P] scans env.step over
// the repeats with no termination check between them, sums their rewards and adds action_repeat to the step counter): the first repeat
// follows the auto-reset prologue, the later ones add their reward to the sum in `reward`, and only the last one runs the episode counter,
// the truncation and the auto-reset; observation, done and metrics are therefore the last repeat's.
#define TM_REP_FIRST 1
#define TM_REP_LAST 2
#define TM_REP_MAX 1024
#define TM_REP(R, first, last) (((R) << 2) | ((first) ? TM_REP_FIRST : 0) | ((last) ? TM_REP_LAST : 0))
#define TM_REP_ONE TM_REP(1, 1, 1)
// `win`: per-(dim, env) partials [2*nu][n] produced by the (env x action-dim)-parallel window kernel, or nullptr to
// compute the window terms inline (lane-per-env path).
// `split`: the observation was written by k_obs and the auto-reset copies are left to k_autoreset (tmjx_hip.hip).
// `fo`: the CALLER's reference frame per env (row-major [n][3 | 4 | nq - 7 | (nbody - 1) * 3 | 3]) in place of the kernel's own gather from the
// resident clip table — compute_tracking_rewards(data, reference_frame, ...) as the reference calls it (reward.py:359-366 with the frame of
// single_clip_tracking.py:223-225); only the inline (lane-per-env) form takes it.
struct TmFrame { const float *pos, *quat, *joints, *bodypos, *angvel; };
TM_DEV void tm_step_post(const DModel &m, EnvRef r, int *r_is, const float *action, float *obs, float *reward, float *done_out,
                         float *trunc_out, float *metrics, const float *win = nullptr, bool split = false, const float *P = nullptr,
                         int rep = TM_REP_ONE, const TmFrame *fo = nullptr) {
  int nu = m.nu, W = m.window, nj = m.nq - 7, nbp = m.nbody - 1;
  int clip = IS(m.i_clip_idx), start = IS(m.i_start_frame), bi = IS(m.i_buffer_index);
  int frame = tm_cur_frame(m, ST(m.s_time, 0), start);
  size_t row = tm_clip_row(m, clip, frame);
  const float *rp = m.clip_pos + row * 3, *rq = m.clip_quat + row * 4, *rj = m.clip_joints + row * nj;
  const float *rb = m.clip_bodypos + row * (size_t)(nbp * 3), *rwv = m.clip_angvel + row * 3;
  if (fo) {
    const size_t e = (size_t)r.e;
    rp = fo->pos + e * 3; rq = fo->quat + e * 4; rj = fo->joints + e * nj; rb = fo->bodypos + e * (size_t)(nbp * 3); rwv = fo->angvel + e * 3;
  }
  const float *w = m.rw;
  // info updates precede the reward call
  float ctrl_sq = 0.f, ctrl_diff = 0.f;
  // (eight actions are loaded before the first of them is stored: the compiler cannot prove that the action rows and the state rows do not
  // overlap, and one load -> store -> load chain per actuator is 38 memory round trips in a row for the lane; same order of summation)
  for (int i0 = 0; i0 < nu; i0 += 8) {
    float av[8];
#pragma unroll
    for (int k = 0; k < 8; k++) av[k] = OUTROW(action, i0 + k < nu ? i0 + k : nu - 1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (i0 + k >= nu) break;
      float a = av[k];
      ST(m.s_prev_ctrl, i0 + k) = a;
      ctrl_sq += a * a;
      float df = a - a;  // prev_ctrl == action at this point (reference quirk); keeps NaN/inf propagation
      ctrl_diff += df * df;
    }
  }
  int bi_old = bi;
  bi = (bi + 1) % W;
  IS(m.i_buffer_index) = bi;
  float pd[3], s = 0.f;
  for (int k = 0; k < 3; k++) { pd[k] = ST(m.s_qpos, k) - rp[k]; s += pd[k] * pd[k]; }
  float pos_reward = w[RW_POS_W] * expf(-w[RW_POS_S] * s);
  float q1[4], q2[4], n1 = 0.f, n2 = 0.f, dt = 0.f;
  for (int k = 0; k < 4; k++) { q1[k] = ST(m.s_qpos, 3 + k); q2[k] = rq[k]; n1 += q1[k] * q1[k]; n2 += q2[k] * q2[k]; }
  n1 = sqrtf(n1); n2 = sqrtf(n2);
  for (int k = 0; k < 4; k++) dt += (q1[k] / n1) * (q2[k] / n2);
  float dist = fminf(1.f, 2.f * dt * dt - 1.f), bq = 0.5f * acosf(dist), quat_distance = bq * bq;
  float quat_reward = w[RW_QUAT_W] * expf(-w[RW_QUAT_S] * quat_distance);
  float joint_distance = 0.f;
  if (P) joint_distance = OUTROW(P, 0);
  else for (int k = 0; k < nj; k++) { float df = ST(m.s_qpos, 7 + k) - rj[k]; joint_distance += df * df; }
  float joint_reward = w[RW_JOINT_W] * expf(-w[RW_JOINT_S] * joint_distance);
  s = 0.f;
  for (int k = 0; k < 3; k++) { float df = ST(m.s_qvel, 3 + k) - rwv[k]; s += df * df; }
  float angvel_reward = w[RW_ANGVEL_W] * expf(-w[RW_ANGVEL_S] * s);
  s = 0.f;
  if (P) s = OUTROW(P, 1);
  else for (int k = 0; k < m.n_body_idx; k++) {
    int i = tm_clampi(m.body_idxs[k], 0, nbp - 1);
    for (int c = 0; c < 3; c++) { float df = ST(m.s_xpos, (1 + i) * 3 + c) - rb[i * 3 + c]; s += df * df; }
  }
  float bodypos_reward = w[RW_BODYPOS_W] * expf(-w[RW_BODYPOS_S] * s);
  s = 0.f;
  if (P) s = OUTROW(P, 2);
  else for (int k = 0; k < m.n_endeff_idx; k++) {
    int i = tm_clampi(m.endeff_idxs[k], 0, nbp - 1);
    for (int c = 0; c < 3; c++) { float df = ST(m.s_xpos, (1 + i) * 3 + c) - rb[i * 3 + c]; s += df * df; }
  }
  float endeff_reward = w[RW_ENDEFF_W] * expf(-w[RW_ENDEFF_S] * s);
  float ctrl_cost = w[RW_CTRL_W] * ctrl_sq, ctrl_diff_cost = w[RW_CTRL_DIFF_W] * ctrl_diff;
  s = 0.f;
  if (P) s = OUTROW(P, 3);
  else for (int i = 6; i < m.nv; i++) s += fabsf(ST(m.s_qvel, i)) * fabsf(ST(m.s_qfrc_actuator, i));
  float energy_cost = w[RW_ENERGY_W] * fminf(s, 50.f);
  float torso_z = ST(m.s_xpos, m.torso_idx * 3 + 2);
  float healthy = torso_z < w[RW_ZLO] ? 0.f : 1.f;
  if (torso_z > w[RW_ZHI]) healthy = 0.f;
  float fall = 1.f - healthy, spd = 0.f;
  for (int k = 0; k < 3; k++) { float x = pd[k] * w[RW_PEN0 + k]; spd += x * x; }
  float too_far = spd > w[RW_TOO_FAR] ? 1.f : 0.f, bad_pose = joint_distance > w[RW_BAD_POSE] ? 1.f : 0.f,
        bad_quat = quat_distance > w[RW_BAD_QUAT] ? 1.f : 0.f;
  // action-variance and jerk costs over the ring buffer (reward.py:314-356)
  float var_sum = 0.f, jerk = 0.f;
  for (int i = 0; i < nu; i++) {
    float vi, ji;
    if (win) { vi = OUTROW(win, i); ji = OUTROW(win, nu + i); }
    else tm_window_dim(m, r, i, bi_old, OUTROW(action, i), vi, ji);
    var_sum += vi; jerk += ji;
  }
  float var_cost = w[RW_VAR_COEFF] * var_sum, jerk_cost = w[RW_JERK_COEFF] * jerk;
  if (!split) tm_get_obs(m, r, clip, frame, obs, true);
  float rew = joint_reward + pos_reward + quat_reward + angvel_reward + bodypos_reward + endeff_reward - ctrl_cost -
              ctrl_diff_cost - energy_cost - var_cost - jerk_cost;
  float done = fmaxf(fmaxf(fall, too_far), fmaxf(bad_pose, bad_quat));
  rew = tm_nan_to_num(rew);
  float nanf_ = P ? fmaxf(OUTROW(P, 4), fmaxf(OUTROW(P, 5), OUTROW(P, 6))) : (tm_state_has_nan(m, r) ? 1.f : 0.f);
  done = fmaxf(done, nanf_);
  OUTROW(metrics, 0) = pos_reward; OUTROW(metrics, 1) = quat_reward; OUTROW(metrics, 2) = joint_reward;
  OUTROW(metrics, 3) = angvel_reward; OUTROW(metrics, 4) = bodypos_reward; OUTROW(metrics, 5) = endeff_reward;
  OUTROW(metrics, 6) = -ctrl_cost; OUTROW(metrics, 7) = -ctrl_diff_cost; OUTROW(metrics, 8) = -energy_cost;
  OUTROW(metrics, 9) = done; OUTROW(metrics, 10) = too_far; OUTROW(metrics, 11) = bad_pose; OUTROW(metrics, 12) = bad_quat;
  OUTROW(metrics, 13) = fall; OUTROW(metrics, 14) = nanf_; OUTROW(metrics, 15) = joint_distance; OUTROW(metrics, 16) = spd;
  OUTROW(metrics, 17) = quat_distance; OUTROW(metrics, 18) = -var_cost; OUTROW(metrics, 19) = -jerk_cost;
  // EpisodeWrapper
  if (!(rep & TM_REP_FIRST)) rew += reward[r.e];
  if (!(rep & TM_REP_LAST)) { reward[r.e] = rew; return; }
  float steps = ST(m.s_steps, 0) + (float)(rep >> 2);
  ST(m.s_steps, 0) = steps;
  bool over = steps >= (float)m.episode_length;
  float trunc = over ? 1.f - done : 0.f;
  if (over) done = 1.f;
  ST(m.s_done, 0) = done;
  reward[r.e] = rew; done_out[r.e] = done; trunc_out[r.e] = trunc;
  // auto-reset: pipeline_state, obs, prev_ctrl <- snapshot taken at reset
  if (done != 0.f && m.auto_reset && !split) {
    for (int i = 0; i < m.nphys; i++) ST(m.s_qpos, i) = ST(m.s_first_phys, i);
    for (int i = 0; i < m.obs_size; i++) OUTROW(obs, i) = ST(m.s_first_obs, i);
    for (int i = 0; i < nu; i++) ST(m.s_prev_ctrl, i) = ST(m.s_first_prev_ctrl, i);
  }
}

// reset_from_clip before pipeline_init: state from the clip frame + injected noise, zeroed task info
TM_DEV void tm_reset_pre(const DModel &m, EnvRef r, int *r_is, int clip, int start, const float *qn, const float *vn) {
  int nj = m.nq - 7;
  size_t row = tm_clip_row(m, clip, start);
  const float *rp = m.clip_pos + row * 3, *rq = m.clip_quat + row * 4, *rj = m.clip_joints + row * nj;
  for (int k = 0; k < 3; k++) ST(m.s_qpos, k) = rp[k] + OUTROW(qn, k);
  for (int k = 0; k < 4; k++) ST(m.s_qpos, 3 + k) = rq[k] + OUTROW(qn, 3 + k);
  for (int k = 0; k < nj; k++) ST(m.s_qpos, 7 + k) = rj[k] + OUTROW(qn, 7 + k);
  for (int i = 0; i < m.nv; i++) { ST(m.s_qvel, i) = OUTROW(vn, i); ST(m.s_warm, i) = 0.f; }
  for (int a = 0; a < m.nu; a++) { ST(m.s_act, a) = 0.f; WS(m.w_ctrl, a) = 0.f; ST(m.s_prev_ctrl, a) = 0.f; ST(m.s_first_prev_ctrl, a) = 0.f; }
  ST(m.s_time, 0) = 0.f; ST(m.s_done, 0) = 0.f; ST(m.s_steps, 0) = 0.f;
  for (int i = 0; i < m.window * m.nu; i++) ST(m.s_action_buffer, i) = 0.f;
  IS(m.i_clip_idx) = clip; IS(m.i_start_frame) = start; IS(m.i_buffer_index) = 0; IS(m.i_nan_count) = 0;
}
// after pipeline_init (mjx.forward): first observation and the auto-reset snapshot
TM_DEV void tm_reset_post(const DModel &m, EnvRef r, int *r_is, float *obs) {
  int clip = IS(m.i_clip_idx), start = IS(m.i_start_frame);
  tm_get_obs(m, r, clip, tm_cur_frame(m, ST(m.s_time, 0), start), obs, false);
  for (int i = 0; i < m.nphys; i++) ST(m.s_first_phys, i) = ST(m.s_qpos, i);
  for (int i = 0; i < m.obs_size; i++) ST(m.s_first_obs, i) = OUTROW(obs, i);
}
