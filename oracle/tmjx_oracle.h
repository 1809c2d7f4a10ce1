/* oracle/tmjx_oracle.h — CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Plain-C restatement of the reference's hot path for parity checking:
 *   physics  : mujoco-mjx 3.3.2 `mjx.step` (third-party, absent from /root/reference;
 *              reached through track_mjx/environment/task/single_clip_tracking.py:219)
 *   env/task : track_mjx/environment/task/{single,multi}_clip_tracking.py, reward.py,
 *              walker/base.py, wrappers.py
 *   learner  : track_mjx/agent/mlp_ppo/losses.py:39-100 (GAE)
 *
 * PARITY UNPINNED: the reference ships no golden vectors and mujoco/jax are not
 * installable here, so the physics part restates MJX's published algorithm from
 * memory.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * Build: `make -C oracle`  ->  oracle/liboracle_f32.so, oracle/liboracle_f64.so
 * The arithmetic type is `real` (float by default, double with -DORACLE_DOUBLE);
 * the C API always exchanges doubles and int32 so one ctypes binding serves both.
 */
#ifndef TMJX_ORACLE_H
#define TMJX_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#if defined(ORACLE_COUNT)
#include "counted_real.hpp"   /* C++ build: `real` counts its own arithmetic (oracle/count_flops.py) */
#elif defined(ORACLE_DOUBLE)
typedef double real;
#else
typedef float real;
#endif

#define O_MAXB 80   /* bodies */
#define O_MAXV 80   /* dofs   */
#define O_MAXQ 80
#define O_MAXU 40
#define O_MAXC 32   /* contact slots */
#define O_MAXEFC (O_MAXV + 4 * O_MAXC)
#define O_MAXW 64   /* action window */
#define O_OBS 1024

typedef struct {
  /* dims */
  int nbody, njnt, nq, nv, nu, ncon, nlim, nefc;
  /* tree */
  int body_parentid[O_MAXB], body_rootid[O_MAXB], body_jntadr[O_MAXB], body_jntnum[O_MAXB];
  int body_dofadr[O_MAXB], body_dofnum[O_MAXB];
  int jnt_type[O_MAXV], jnt_bodyid[O_MAXV], jnt_qposadr[O_MAXV], jnt_dofadr[O_MAXV], jnt_limited[O_MAXV];
  int dof_bodyid[O_MAXV], dof_jntid[O_MAXV], dof_parentid[O_MAXV];
  int lim_jnt[O_MAXV];
  real body_pos[O_MAXB][3], body_quat[O_MAXB][4], body_mass[O_MAXB], body_ipos[O_MAXB][3];
  real body_iquat[O_MAXB][4], body_inertia[O_MAXB][3], body_invweight0[O_MAXB][2];
  real jnt_pos[O_MAXV][3], jnt_axis[O_MAXV][3], jnt_range[O_MAXV][2], jnt_stiffness[O_MAXV];
  real jnt_solref[O_MAXV][2], jnt_solimp[O_MAXV][5], jnt_margin[O_MAXV];
  real qpos0[O_MAXQ], qpos_spring[O_MAXQ], dof_damping[O_MAXV], dof_armature[O_MAXV], dof_invweight0[O_MAXV];
  real act_moment[O_MAXU][O_MAXV], act_gain[O_MAXU], act_tau[O_MAXU], act_ctrlrange[O_MAXU][2];
  real gravity[3], meaninertia;
  /* contacts (static slots) */
  int con_type[O_MAXC], con_sub[O_MAXC], con_body1[O_MAXC], con_body2[O_MAXC], con_geom1[O_MAXC], con_geom2[O_MAXC];
  real con_friction[O_MAXC][3], con_solref[O_MAXC][2], con_solimp[O_MAXC][5];
  real con_g1_pos[O_MAXC][3], con_g1_quat[O_MAXC][4], con_g2_pos[O_MAXC][3], con_g2_quat[O_MAXC][4], con_g2_size[O_MAXC][3];
  /* options */
  real timestep, tolerance, ls_tolerance, impratio;
  int iterations, ls_iterations, n_frames;
  /* env / task config */
  int mocap_hz, clip_length, traj_length, window, torso_idx, episode_length, auto_reset;
  int action_repeat;   /* brax EpisodeWrapper's repeat count (wrappers.py:43); 1 unless oracle_set_action_repeat changed it */
  int n_joint_idx, n_body_idx, n_endeff_idx;
  int joint_idxs[O_MAXV], body_idxs[O_MAXB], endeff_idxs[16];
  real rw[32];
  /* clips (float32 tables as uploaded; gather must be bit exact) */
  int n_clips, n_frames_clip;
  float *clip_pos, *clip_quat, *clip_joints, *clip_bodypos, *clip_angvel;
} OModel;

typedef struct {
  real qpos[O_MAXQ], qvel[O_MAXV], act[O_MAXU], ctrl[O_MAXU], qacc_warmstart[O_MAXV], time;
  real xpos[O_MAXB][3], xquat[O_MAXB][4], xmat[O_MAXB][9], xipos[O_MAXB][3], ximat[O_MAXB][9];
  real xanchor[O_MAXV][3], xaxis[O_MAXV][3];
  real subtree_com[O_MAXB][3], cinert[O_MAXB][10], cdof[O_MAXV][6], crb[O_MAXB][10];
  real qM[O_MAXV][O_MAXV], qLD[O_MAXV][O_MAXV], qLDh[O_MAXV][O_MAXV];
  real cvel[O_MAXB][6], cdof_dot[O_MAXV][6];
  real qfrc_bias[O_MAXV], qfrc_passive[O_MAXV], act_dot[O_MAXU], actuator_force[O_MAXU];
  real qfrc_actuator[O_MAXV], qfrc_smooth[O_MAXV], qacc_smooth[O_MAXV];
  real con_dist[O_MAXC], con_pos[O_MAXC][3], con_frame[O_MAXC][9];
  real efc_J[O_MAXEFC][O_MAXV], efc_D[O_MAXEFC], efc_aref[O_MAXEFC], efc_pos[O_MAXEFC], efc_force[O_MAXEFC];
  real qfrc_constraint[O_MAXV], qacc[O_MAXV];
  int solver_niter, ls_total;
} OData;

typedef struct {
  OData d;
  int clip_idx, start_frame, buffer_index;
  real prev_ctrl[O_MAXU], action_buffer[O_MAXW][O_MAXU];
  real steps, truncation;
  real obs[O_OBS], reward, done, metrics[20];
  /* auto-reset snapshot (wrappers.py:93-95) */
  real first_obs[O_OBS], first_prev_ctrl[O_MAXU];
  OData first_d;
  /* the CALLER's reference frame for the reward terms (compute_tracking_rewards(data, reference_frame, ...), reward.py:359-366): when fo_pos is
   * non-NULL the five leaves of ONE frame (3 | 4 | nq-7 | (nbody-1)*3 | 3 floats) replace the env's own gather from the clip table */
  const float *fo_pos, *fo_quat, *fo_joints, *fo_bodypos, *fo_angvel;
} OEnv;

#ifdef __cplusplus
extern "C" {
#endif
OModel *oracle_model_create(const void *blob, size_t nbytes);
void oracle_model_destroy(OModel *m);
int oracle_set_clips(OModel *m, const float *pos, const float *quat, const float *joints,
                     const float *bodypos, const float *angvel, int n_clips, int n_frames);
size_t oracle_sizeof_data(void);
size_t oracle_sizeof_env(void);
void oracle_data_init(const OModel *m, OData *d, const double *qpos, const double *qvel);
void oracle_forward(const OModel *m, OData *d);
void oracle_step(const OModel *m, OData *d);
void oracle_set_ctrl(const OModel *m, OData *d, const double *ctrl);
int oracle_data_get(const OModel *m, const OData *d, const char *name, double *out, int cap);
int oracle_data_set(const OModel *m, OData *d, const char *name, const double *in, int n);
/* env */
void oracle_env_reset(const OModel *m, OEnv *e, int clip_idx, int start_frame,
                      const double *qpos_noise, const double *qvel_noise);
void oracle_env_step(const OModel *m, OEnv *e, const double *action);
/* do_physics = 0: everything except pipeline_step (K3 alone: frame gather, rewards, obs, wrappers) */
void oracle_env_step_ex(const OModel *m, OEnv *e, const double *action, int do_physics);
void oracle_set_action_repeat(OModel *m, int action_repeat);
int oracle_env_set(const OModel *m, OEnv *e, const char *name, const double *in, int n);
int oracle_env_get(const OModel *m, const OEnv *e, const char *name, double *out, int cap);
int oracle_obs_size(const OModel *m);
void oracle_env_step_batch(const OModel *m, OEnv *envs, int n, const double *actions, int nthreads);
/* learner */
void oracle_gae(const double *truncation, const double *termination, const double *rewards,
                const double *values, const double *bootstrap, double lambda_, double discount,
                double *vs, double *adv, int T, int B);
#ifdef __cplusplus
}
#endif
#endif
