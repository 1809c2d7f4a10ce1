#!/usr/bin/env python3
"""Generates tests/golden/task_golden.npz: known-answer vectors for the track_mjx-owned maths of the hot path.

The reference ships no tests or golden vectors and cannot be imported here (jax/brax/mujoco are not installed), so the
expected values come from this independent float32 numpy restatement, written line by line from the reference sources
cited below (SURVEY.md Appendix B lists the cases).  The C oracle (oracle/tmjx_oracle_env.c) and the HIP kernels are
both checked against these vectors.

  reward.py:57-77    _bounded_quat_dist          reward.py:80-216  pos/quat/joint/angvel/bodypos/endeff rewards
  reward.py:219-260  ctrl / ctrl_diff / energy   reward.py:263-311 health + penalty terms
  reward.py:314-356  action variance / jerk      single_clip_tracking.py:322-454 obs + frame index
  walker/base.py:170-258 egocentric maths (brax.math.rotate / relative_quat)   losses.py:39-100 GAE
Two fixture files, one per shipped rodent configuration (track_mjx_amd/config.py: NAMED_CONFIGS): task_golden.npz = rodent-full-clips.yaml,
task_golden_sps.npz = rodent-sps-per-actor.yaml (5 substeps per control step => 0.01 s per step, penalty scale [1, 1, 0.2], RewardConfig's
default var / jerk coefficients, energy term off).
Run: python tests/golden/make_golden.py   (deterministic; commit the .npz files)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from track_mjx_amd import clips as _clips, config as _config, walker as _walker  # noqa: E402

f32 = np.float32


def rotate(v, q):  # brax.math.rotate(vec, quat)
    s, u = q[0], q[1:]
    r = f32(2) * (np.dot(u, v) * u) + (s * s - np.dot(u, u)) * v
    return (r + f32(2) * s * np.cross(u, v)).astype(f32)


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]], dtype=f32)


def relative_quat(q1, q2):  # brax.math.relative_quat: quat_mul(q2, quat_inv(q1))
    return quat_mul(q2, q1 * np.array([1, -1, -1, -1], dtype=f32))


def bounded_quat_dist(a, b):
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    d = f32(2) * np.dot(a, b) ** 2 - f32(1)
    return f32(0.5) * np.arccos(np.minimum(f32(1), d))


def cur_frame(time, hz, start):
    return int(np.floor(f32(f32(time) * f32(hz)) + f32(start)))


def step_terms(case, clip, w, cfg):
    """Everything MultiClipTracking.step computes after pipeline_step, for one env."""
    rw = cfg["env_config"]["reward_weights"]
    qpos, qvel, xpos, xmat_torso, qfa = case["qpos"], case["qvel"], case["xpos"].reshape(68, 3), case["xmat_torso"], case["qfrc_actuator"]
    action, buf, bidx = case["action"], case["action_buffer"].reshape(50, 38).copy(), int(case["buffer_index"])
    c, frame = int(case["clip_idx"]), cur_frame(case["time"], 50, case["start_frame"])
    frame_c = min(max(frame, 0), 249)
    ref = {k: getattr(clip, k)[c, frame_c] for k in ("position", "quaternion", "joints", "body_positions", "angular_velocity")}
    # info updates precede the reward call: prev_ctrl = action (=> ctrl_diff_cost 0), buffer write, index advance
    buf[bidx] = action; bidx = (bidx + 1) % 50
    pos_distance = qpos[:3] - ref["position"]
    pos_reward = f32(rw["pos_reward_weight"]) * np.exp(-f32(rw["pos_reward_exp_scale"]) * np.sum(pos_distance ** 2))
    quat_distance = np.sum(bounded_quat_dist(qpos[3:7], ref["quaternion"]) ** 2)
    quat_reward = f32(rw["quat_reward_weight"]) * np.exp(-f32(rw["quat_reward_exp_scale"]) * quat_distance)
    joint_distance = np.sum((qpos[7:] - ref["joints"]) ** 2)
    joint_reward = f32(rw["joint_reward_weight"]) * np.exp(-f32(rw["joint_reward_exp_scale"]) * joint_distance)
    angvel_reward = f32(rw["angvel_reward_weight"]) * np.exp(-f32(rw["angvel_reward_exp_scale"]) * np.sum((qvel[3:6] - ref["angular_velocity"]) ** 2))
    x1 = xpos[1:]
    bidx_c = np.minimum(w.body_idxs, 66); eidx_c = np.minimum(w.endeff_idxs, 66)   # XLA gather clamps OOB index 67 -> 66
    bodypos_reward = f32(rw["bodypos_reward_weight"]) * np.exp(-f32(rw["bodypos_reward_exp_scale"]) * np.sum((x1[bidx_c] - ref["body_positions"][bidx_c]) ** 2))
    endeff_reward = f32(rw["endeff_reward_weight"]) * np.exp(-f32(rw["endeff_reward_exp_scale"]) * np.sum((x1[eidx_c] - ref["body_positions"][eidx_c]) ** 2))
    ctrl_cost = f32(rw["ctrl_cost_weight"]) * np.sum(action ** 2)
    ctrl_diff_cost = f32(rw["ctrl_diff_cost_weight"]) * np.sum((action - action) ** 2)
    energy_cost = f32(rw["energy_cost_weight"]) * np.minimum(np.sum(np.abs(qvel[6:]) * np.abs(qfa[6:])), f32(50))
    torso_z = xpos[w.torso_idx][2]
    healthy = 0.0 if torso_z < f32(rw["healthy_z_range"][0]) else 1.0
    if torso_z > f32(rw["healthy_z_range"][1]):
        healthy = 0.0
    fall = 1.0 - healthy
    spd = np.sum((pos_distance * np.array(rw["penalty_pos_distance_scale"], dtype=f32)) ** 2)
    too_far = float(spd > f32(rw["too_far_dist"])); bad_pose = float(joint_distance > f32(rw["bad_pose_dist"])); bad_quat = float(quat_distance > f32(rw["bad_quat_dist"]))
    mean_act = buf.mean(0); var_cost = f32(rw.get("var_coeff", 5e-2)) * np.sum(((buf - mean_act) ** 2).mean(0))   # reward.py:52: default 5e-2
    ordered = np.concatenate([buf, buf], 0)[bidx:bidx + 50]
    jerks = ordered[2:] - 2 * ordered[1:-1] + ordered[:-2]
    jerk_cost = f32(rw.get("jerk_coeff", 5e-4)) * np.sum(jerks ** 2)
    reward = (joint_reward + pos_reward + quat_reward + angvel_reward + bodypos_reward + endeff_reward - ctrl_cost - ctrl_diff_cost
              - energy_cost - var_cost - jerk_cost)
    done = max(fall, too_far, bad_pose, bad_quat)
    # observation
    start = min(max(frame + 1, 0), 250 - 5)
    quat = qpos[3:7]
    o = []
    for t in range(5):
        o.append(rotate(clip.position[c, start + t] - qpos[:3], quat))
    for t in range(5):
        o.append(relative_quat(clip.quaternion[c, start + t], quat))
    for t in range(5):
        o.append((clip.joints[c, start + t] - qpos[7:])[w.joint_idxs - 1])
    for t in range(5):
        d = (clip.body_positions[c, start + t] - x1)[bidx_c]
        o.append(np.concatenate([rotate(v, quat) for v in d]))
    X = xmat_torso.reshape(3, 3)
    app = (xpos[w.endeff_idxs] - xpos[w.torso_idx]) @ X
    o += [qpos[7:], qvel[6:], qfa, np.array([torso_z]), xmat_torso[6:], app.ravel()]
    obs = np.concatenate([np.asarray(x, dtype=f32).ravel() for x in o])
    metrics = np.array([pos_reward, quat_reward, joint_reward, angvel_reward, bodypos_reward, endeff_reward, -ctrl_cost, -ctrl_diff_cost,
                        -energy_cost, done, too_far, bad_pose, bad_quat, fall, 0.0, joint_distance, spd, quat_distance, -var_cost, -jerk_cost], dtype=f32)
    return dict(reward=f32(reward), done=f32(done), obs=obs.astype(f32), metrics=metrics, frame=np.int32(frame), buffer_index=np.int32(bidx), action_buffer=buf.ravel())


def gae_numpy(trunc, term, rew, val, boot, lam, disc):
    T, B = rew.shape
    tm = 1 - trunc
    v1 = np.concatenate([val[1:], boot[None]], 0)
    deltas = (rew + disc * (1 - term) * v1 - val) * tm
    acc = np.zeros(B, dtype=f32); out = np.zeros((T, B), dtype=f32)
    for t in range(T - 1, -1, -1):
        acc = deltas[t] + disc * (1 - term[t]) * tm[t] * lam * acc
        out[t] = acc
    vs = out + val
    vs1 = np.concatenate([vs[1:], boot[None]], 0)
    adv = (rew + disc * (1 - term) * vs1 - val) * tm
    return vs.astype(f32), adv.astype(f32)


def main(config="rodent-full-clips", fname="task_golden.npz", seed=2024):
    cfg = _config.named_config(config)
    ea = cfg["env_config"]["env_args"]
    dt = f32(ea["mj_model_timestep"] * ea["physics_steps_per_control_step"])     # one control step
    w = _walker.Rodent(**cfg["walker_config"])
    clip = _clips.make_synthetic_clips(w.model, 3, seed=123)
    rng = np.random.default_rng(seed)
    out = {}
    n = 24
    keys_in = ("qpos", "qvel", "xpos", "xmat_torso", "qfrc_actuator", "action", "action_buffer", "buffer_index", "clip_idx", "start_frame", "time")
    cases = []
    for i in range(n):
        c = dict(qpos=(rng.normal(size=74) * 0.2).astype(f32), qvel=rng.normal(size=73).astype(f32), xpos=(rng.normal(size=204) * 0.1).astype(f32),
                 xmat_torso=rng.normal(size=9).astype(f32), qfrc_actuator=rng.normal(size=73).astype(f32), action=rng.uniform(-1, 1, 38).astype(f32),
                 action_buffer=rng.uniform(-1, 1, 1900).astype(f32), buffer_index=np.int32([0, 1, 49, 17][i % 4]), clip_idx=np.int32(i % 3),
                 start_frame=np.int32(rng.integers(0, 44)), time=f32(rng.integers(0, 195)) * dt)
        if i < 6:   # near-reference states: rewards away from 0, health inside the band, edge cases of the thresholds
            fr = cur_frame(c["time"], 50, c["start_frame"])
            c["qpos"] = np.concatenate([clip.position[c["clip_idx"], fr], clip.quaternion[c["clip_idx"], fr], clip.joints[c["clip_idx"], fr]]).astype(f32) + (rng.normal(size=74) * 0.01).astype(f32)
            c["xpos"] = np.concatenate([[0, 0, 0], clip.body_positions[c["clip_idx"], fr].ravel()]).astype(f32) + (rng.normal(size=204) * 0.002).astype(f32)
            c["xpos"][3 * 3 + 2] = [0.03, 0.0325, 0.5, 0.51, 0.2, 0.1][i]
        if i == 6:
            c["time"] = f32(4.9); c["start_frame"] = np.int32(43)   # frame + 1 + 5 > 250: dynamic_slice start clamp
        cases.append(c)
    for k in keys_in:
        out["in_" + k] = np.stack([np.asarray(c[k]) for c in cases])
    res = [step_terms(c, clip, w, cfg) for c in cases]
    for k in res[0]:
        out["out_" + k] = np.stack([r[k] for r in res])
    # frame-index table: steps 1..195 x start 0..43, time accumulated by 10 fp32 adds of 0.002 per step (FMA-sensitive)
    t = f32(0); table = np.zeros((195, 44), dtype=np.int32); times = np.zeros(195, dtype=f32)
    for s in range(195):
        for _ in range(ea["physics_steps_per_control_step"]):
            t = f32(t + f32(0.002))
        times[s] = t
        table[s] = [cur_frame(t, 50, st) for st in range(44)]
    out["frame_times"], out["frame_table"] = times, table
    # quaternion distance known answers
    out["bqd_cases"] = np.array([[1, 0, 0, 0, 1, 0, 0, 0], [np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4), 1, 0, 0, 0], [0.3, 0.4, 0.5, 0.6, -0.3, -0.4, -0.5, -0.6],
                                 [2, 0, 0, 0, 0, 0, 3, 0]], dtype=f32)
    out["bqd_expect"] = np.array([bounded_quat_dist(r[:4], r[4:]) for r in out["bqd_cases"]], dtype=f32)
    # GAE: hand-sized case with one termination and one truncation + a random one
    T, B = 20, 16
    trunc = (rng.random((T, B)) < 0.1).astype(f32); term = ((rng.random((T, B)) < 0.1) * (1 - trunc)).astype(f32)
    rew, val, boot = rng.normal(size=(T, B)).astype(f32), rng.normal(size=(T, B)).astype(f32), rng.normal(size=B).astype(f32)
    vs, adv = gae_numpy(trunc, term, rew, val, boot, f32(0.95), f32(0.98))
    out.update(gae_trunc=trunc, gae_term=term, gae_rew=rew, gae_val=val, gae_boot=boot, gae_vs=vs, gae_adv=adv)
    np.savez_compressed(Path(__file__).with_name(fname), **out)
    print(f"wrote {fname}:", {k: v.shape for k, v in out.items() if k.startswith("out_")})


if __name__ == "__main__":
    main()
    main("rodent-sps-per-actor", "task_golden_sps.npz", seed=2025)
