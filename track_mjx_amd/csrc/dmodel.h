// csrc/dmodel.h — device-resident model constants and buffer layout of the hot path.
//
// One DModel lives in device memory per handle; kernels read it through wave-uniform indices so the
// compiler can use scalar (s_load) loads.  All floats are fp32 (MJX narrows MuJoCo's float64 model the
// same way: mjx.put_model, reference call site track_mjx/environment/task/single_clip_tracking.py:91).
#pragma once
#include <stdint.h>

#define TM_MAXB 72    // bodies
#define TM_MAXV 76    // dofs
#define TM_MAXQ 76
#define TM_MAXU 40
#define TM_MAXC 32    // contact slots
#define TM_MAXG 8     // contact groups (distinct paw bodies)
#define TM_MAXIDX 40
#define TM_NMETRIC 20

struct DModel {
  int nbody, njnt, nq, nv, nu, ncon, nlim, nefc, ngroup, nnz;
  int root_body;  // the single moving tree's root (walker)
  // kinematic tree
  int body_parentid[TM_MAXB], body_jntadr[TM_MAXB], body_jntnum[TM_MAXB], body_dofadr[TM_MAXB], body_dofnum[TM_MAXB];
  int body_moving[TM_MAXB];  // 1 if rootid == root_body
  int jnt_type[TM_MAXV], jnt_bodyid[TM_MAXV], jnt_qposadr[TM_MAXV], jnt_dofadr[TM_MAXV];
  int dof_bodyid[TM_MAXV], dof_parentid[TM_MAXV], dof_Madr[TM_MAXV], dof_depth[TM_MAXV];
  int lim_jnt[TM_MAXV];
  float body_pos[TM_MAXB][3], body_quat[TM_MAXB][4], body_mass[TM_MAXB], body_ipos[TM_MAXB][3], body_iquat[TM_MAXB][4],
      body_inertia[TM_MAXB][3];
  float jnt_pos[TM_MAXV][3], jnt_axis[TM_MAXV][3], jnt_range[TM_MAXV][2], jnt_stiffness[TM_MAXV];
  float jnt_solref[TM_MAXV][2], jnt_solimp[TM_MAXV][5], jnt_margin[TM_MAXV];
  float qpos0[TM_MAXQ], qpos_spring[TM_MAXQ], dof_damping[TM_MAXV], dof_armature[TM_MAXV], dof_invweight0[TM_MAXV];
  // actuators: sparse moment (joint transmissions are one-hot, fixed tendons a few entries)
  int act_madr[TM_MAXU + 1], act_mdof[128];
  float act_mval[128], act_gain[TM_MAXU], act_tau[TM_MAXU], act_ctrlrange[TM_MAXU][2];
  float gravity[3], meaninertia;
  // contact slots in MJX order; slots of one paw body are contiguous ("group")
  int con_type[TM_MAXC], con_sub[TM_MAXC], con_body1[TM_MAXC], con_body2[TM_MAXC];
  float con_mu[TM_MAXC], con_solref[TM_MAXC][2], con_solimp[TM_MAXC][5], con_invweight[TM_MAXC];
  float con_g1_pos[TM_MAXC][3], con_g1_quat[TM_MAXC][4], con_g2_pos[TM_MAXC][3], con_g2_quat[TM_MAXC][4], con_g2_size[TM_MAXC][3];
  int grp_body[TM_MAXG], grp_lastdof[TM_MAXG], grp_start[TM_MAXG], grp_count[TM_MAXG];
  // options
  float timestep, tolerance, ls_tolerance, impratio;
  int iterations, ls_iterations, n_frames;
  // task
  int mocap_hz, clip_length, traj_length, window, torso_idx, episode_length, auto_reset;
  int n_joint_idx, n_body_idx, n_endeff_idx;
  int joint_idxs[TM_MAXIDX], body_idxs[TM_MAXIDX], endeff_idxs[TM_MAXIDX];
  float rw[32];
  int obs_size, ref_obs_size;
  // clip table (device pointers, float32)
  int n_clips, n_frames_clip;
  const float *clip_pos, *clip_quat, *clip_joints, *clip_bodypos, *clip_angvel;
  // ---- row offsets: float state buffer
  int s_qpos, s_qvel, s_act, s_warm, s_time, s_xpos, s_xmat_torso, s_qfrc_actuator, s_prev_ctrl, s_action_buffer,
      s_done, s_steps, s_first_phys, s_first_obs, s_first_prev_ctrl, s_rows;
  int nphys;  // qpos..time rows (contiguous)
  // int state buffer
  int i_clip_idx, i_start_frame, i_buffer_index, i_nan_count, i_rows;
  // ---- row offsets: workspace
  int w_ctrl, w_xquat, w_xanchor, w_xaxis, w_xipos, w_cinert, w_cdof, w_crb, w_M, w_LD, w_Dinv, w_cvel, w_cdof_dot, w_cacc,
      w_cfrc, w_qfrc_smooth, w_qacc_smooth, w_act_dot, w_con_dist, w_con_off, w_con_frame, w_efc_D, w_efc_aref,
      w_efc_Jaref, w_efc_jv, w_lim_sign, w_qacc, w_Ma, w_grad, w_Mgrad, w_search, w_mv, w_qfrc_constraint, w_tmp,
      w_efc_force, w_com, w_solver_stats, w_efc_in, w_rows;

  // ================= tables and LDS map of the wave-per-env physics kernel (csrc/wave_physics.h) =================
  // bodies
  int scan_parent[TM_MAXB];   // ancestor pointer for the transform scan (-1: transform is already absolute)
  int body_lastdof[TM_MAXB];  // last dof on the path world -> body (own dofs included), -1 if none
  int child_adr[TM_MAXB + 1], child_ids[TM_MAXB];
  int nlevel, lvl_adr[TM_MAXB + 1], lvl_parents[TM_MAXB];  // level L (deepest first): parents that own children
  int nround_body, nround_dof;
  // dofs
  int dof_jntid[TM_MAXV], dof_vpar[TM_MAXV], dof_ndesc[TM_MAXV], dof_limrow[TM_MAXV], dof_freetrans[TM_MAXV];
  int dof_act_adr[TM_MAXV + 1], dof_act_id[128];
  float dof_act_coef[128], dof_stiffness[TM_MAXV], dof_qspring[TM_MAXV];
  int dof_qposadr[TM_MAXV];
  int dof_grp_adr[TM_MAXV + 1], dof_grp_ids[TM_MAXV * TM_MAXG];
  int con_grp[TM_MAXC];
  // J^T f by wrench SUBSETS: every dof feels the summed wrench of the paw bodies below it; dofs with the same set of paw
  // groups share one sum.  wsub_cmask[s]: contacts (bit c of word c / 32) of subset s; dof_wsub[i]: subset of dof i or -1
  int n_wsub, dof_wsub[TM_MAXV];
  unsigned wsub_cmask[TM_MAXV][2];
  // tree-sparse rows: ancestor tables, one entry per stored non-zero (entry k of row i = k-th ancestor of dof i)
  uint8_t anc_dof[1280];
  uint16_t anc_Madr[1280];
  // column access: for dof i, entries (descendant k ascending) = offset of M(k,i) inside the sparse storage
  int dof_coladr[TM_MAXV + 1];
  uint16_t col_off[1280];
  float total_mass;
  // per-dof packed words kept in LDS: [2i] = Madr | depth << 16 ; [2i+1] = chain_start | (jump + 1) << 8
  // (ancestors of dof i: i-1 .. chain_start, then jump, jump-1, .. 0 — checked on the host)
  int tdof[TM_MAXV * 2];
  // lean (rodent chain) kernel: ONE packed word per dof, kept in two registers of the lane that owns dofs lane / lane + 64 (the table above took
  // 146 words of LDS per env): Mend = Madr + depth (11 bits) | depth << 11 (6) | in-chain run i - chain_start << 17 (5) | (wrench subset + 1)
  // << 22 (4); jump + 1 = depth - run, and the limit row of dof i is i - 6 (checked on the host: rodent_chains_match).  gpack[g] = last dof of paw
  // group g (0xff: none) | its depth << 8 | its run << 16: what J v needs of the group's dof (one LDS word per group)
  int tpack[TM_MAXV], gpack[TM_MAXG];
  int body_nsub[TM_MAXB];  // subtree size (bodies are numbered depth-first: subtree = [b, b + nsub))
  // subtree sums (wave kernel): a RUN is a maximal chain b, b+1, .. with parent[b+1] == b.  After the per-run suffix sums,
  // every branch body p (more than one child), taken in DESCENDING order, adds the finished sums of its non-first children
  // (run heads fix_child[fix_adr[f] .. fix_adr[f+1])) to the bodies fix_r0[f] .. fix_p[f] of its own run.
  int n_fix, fix_p[TM_MAXB], fix_r0[TM_MAXB], fix_adr[TM_MAXB + 1], fix_child[TM_MAXB];
  // Flat per-lane records of the wave kernel (round 4).  A dependent chain of per-lane model reads (body -> its joint -> the joint's qpos0 ..)
  // costs about a thousand cycles PER LEVEL next to eleven other waves; everything a lane needs for one body / one dof sits in ONE record
  // that it loads in one level.  body_kin[b] = { body_pos 3 | body_quat 4 | type of the body's FIRST joint (-1: none) | its qposadr | jnt_pos 3 |
  // jnt_axis 3 | qpos0[qposadr] } (ints stored as their bits); bodies with more than one joint take the tables for the others.
  float body_kin[TM_MAXB][16];
  // dof_kin[i] = { type of the dof's joint | body of that joint | parent of that body | dof index inside the joint } (cdof stage);
  // dof_dyn[i] = { body of the dof | armature | damping | stiffness | qposadr | spring reference | first actuator entry | entries end }
  int dof_kin[TM_MAXV][4];
  float dof_dyn[TM_MAXV][8];
  // dof_lim[i] (hinge dofs with a limit) = { qposadr | range lo | range hi | margin | solref 2 | solimp 5 | dof_invweight0 }: the limit row of
  // dof i in one read (make_constraint: row -> dof -> joint -> fields was three levels)
  float dof_lim[TM_MAXV][12];
  float dof_act_gain[128];      // act_gain[dof_act_id[e]] next to dof_act_coef[e]: one level fewer in the dof's actuator gather
  int lds_floats;
};

enum { RW_TOO_FAR, RW_BAD_POSE, RW_BAD_QUAT, RW_CTRL_W, RW_CTRL_DIFF_W, RW_ENERGY_W, RW_POS_W, RW_QUAT_W, RW_JOINT_W,
       RW_ANGVEL_W, RW_BODYPOS_W, RW_ENDEFF_W, RW_ZLO, RW_ZHI, RW_POS_S, RW_QUAT_S, RW_JOINT_S, RW_ANGVEL_S,
       RW_BODYPOS_S, RW_ENDEFF_S, RW_PEN0, RW_PEN1, RW_PEN2, RW_VAR_COEFF, RW_JERK_COEFF };
