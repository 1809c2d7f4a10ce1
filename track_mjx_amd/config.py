"""Configuration of the hot path.

Same key names and nesting as the reference's hydra config
(reference: track_mjx/config/rodent-full-clips.yaml:1-176); hydra/omegaconf are not
available, so plain dicts + PyYAML + `key=value` overrides (reference CLI:
track_mjx/train.py:56, README.md:91).
"""
from __future__ import annotations

import copy
from typing import Any

import numpy as np
import yaml

# order of the float vector handed to the C-ABI as blob entry "reward_f"
REWARD_F = [
    "too_far_dist", "bad_pose_dist", "bad_quat_dist", "ctrl_cost_weight", "ctrl_diff_cost_weight",
    "energy_cost_weight", "pos_reward_weight", "quat_reward_weight", "joint_reward_weight",
    "angvel_reward_weight", "bodypos_reward_weight", "endeff_reward_weight", "healthy_z_lo", "healthy_z_hi",
    "pos_reward_exp_scale", "quat_reward_exp_scale", "joint_reward_exp_scale", "angvel_reward_exp_scale",
    "bodypos_reward_exp_scale", "endeff_reward_exp_scale", "pen0", "pen1", "pen2", "var_coeff", "jerk_coeff",
]

_JOINTS = """vertebra_1_extend hip_L_supinate hip_L_abduct hip_L_extend knee_L ankle_L toe_L hip_R_supinate
hip_R_abduct hip_R_extend knee_R ankle_R toe_R vertebra_C11_extend vertebra_cervical_1_bend vertebra_axis_twist
atlas mandible scapula_L_supinate scapula_L_abduct scapula_L_extend shoulder_L shoulder_sup_L elbow_L wrist_L
scapula_R_supinate scapula_R_abduct scapula_R_extend shoulder_R shoulder_sup_R elbow_R wrist_R finger_R""".split()
_BODIES = """torso pelvis upper_leg_L lower_leg_L foot_L upper_leg_R lower_leg_R foot_R skull jaw scapula_L
upper_arm_L lower_arm_L finger_L scapula_R upper_arm_R lower_arm_R finger_R""".split()
_ENDEFF = "foot_L foot_R hand_L hand_R skull".split()


def default_config() -> dict[str, Any]:
    """The `rodent-full-clips` configuration (values: rodent-full-clips.yaml)."""
    return {
        "data_path": "synthetic",
        "env_config": {
            "env_name": "rodent_multi_clip",
            "env_args": {
                "solver": "cg", "iterations": 5, "ls_iterations": 5,
                "physics_steps_per_control_step": 10, "reset_noise_scale": 1e-3,
                "mj_model_timestep": 0.002, "mocap_hz": 50,
            },
            "reward_weights": {
                "too_far_dist": 0.01, "bad_pose_dist": 20, "bad_quat_dist": 1,
                "ctrl_cost_weight": 0.02, "ctrl_diff_cost_weight": 0.02, "energy_cost_weight": 0.01,
                "pos_reward_weight": 1.0, "quat_reward_weight": 1.0, "joint_reward_weight": 1.0,
                "angvel_reward_weight": 0.0, "bodypos_reward_weight": 0.0, "endeff_reward_weight": 1.0,
                "healthy_z_range": [0.0325, 0.5],
                "pos_reward_exp_scale": 400.0, "quat_reward_exp_scale": 4.0, "joint_reward_exp_scale": 0.25,
                "angvel_reward_exp_scale": 0.5, "bodypos_reward_exp_scale": 8.0, "endeff_reward_exp_scale": 500.0,
                "var_window_size": 50, "var_coeff": 5e-3, "jerk_coeff": 5e-4,
                "penalty_pos_distance_scale": [1.0, 1.0, 0.5],
            },
        },
        "reference_config": {"clip_length": 250, "random_init_range": 50, "traj_length": 5},
        "network_config": {
            "arch_name": "intention",
            "encoder_layer_sizes": [1024, 512, 512, 512, 512],
            "decoder_layer_sizes": [512, 512, 512, 256, 256],
            "critic_layer_sizes": [512, 512, 512, 512, 512, 256],
            "intention_size": 60, "kl_schedule": True, "kl_weight": 1e-1,
        },
        "train_setup": {
            "train_subset_ratio": 0.8, "eval_every": 10_000_000, "episode_length": 200, "reset_every": 10_000_000,
            "train_config": {
                "num_envs": 4096, "num_timesteps": 1_500_000_000, "batch_size": 1024, "num_minibatches": 16,
                "num_updates_per_batch": 4, "learning_rate": 1e-4, "clipping_epsilon": 0.2, "discounting": 0.98,
                "action_repeat": 1, "entropy_cost": 1e-2, "reward_scaling": 1, "normalize_observations": True,
                "unroll_length": 20, "seed": 0, "get_activation": False, "use_lstm": False,
                "deterministic_eval": True,
            },
        },
        "walker_config": {
            "torque_actuators": True, "rescale_factor": 0.9,
            "joint_names": list(_JOINTS), "body_names": list(_BODIES), "end_eff_names": list(_ENDEFF),
        },
    }


def sps_per_actor_config() -> dict[str, Any]:
    """The reference's other shipped rodent configuration (values: rodent-sps-per-actor.yaml:11-17,19-39,45-52,67-82): CG 4 / 4 — also
    the constructor defaults of multi_clip_tracking.py:16-32 —, 5 physics substeps per control step, `penalty_pos_distance_scale`
    [1, 1, 0.2], 512 x 3 nets, and NO `var_*` / `jerk_coeff` keys, i.e. RewardConfig's defaults (reward.py:51-53: window 50, var_coeff
    5e-2, jerk_coeff 5e-4).  That file also omits `energy_cost_weight`, which the reference's dataclass requires (reward.py:27-28): it is
    given as 0.0 here (term off).  The walker stays the shipped blob's (torque actuators, rescale 0.9): the file's position-actuator /
    0.8 walker needs a recompile with tools/compile_model.py and the affine-bias actuator path, which is not built."""
    cfg = default_config()
    cfg["env_config"]["env_args"].update(iterations=4, ls_iterations=4, physics_steps_per_control_step=5)
    rw = cfg["env_config"]["reward_weights"]
    for k in ("var_window_size", "var_coeff", "jerk_coeff"):
        del rw[k]
    rw.update(energy_cost_weight=0.0, penalty_pos_distance_scale=[1.0, 1.0, 0.2])
    cfg["network_config"].update(encoder_layer_sizes=[512, 512, 512], decoder_layer_sizes=[512, 512, 512], critic_layer_sizes=[512, 512, 512],
                                 kl_weight=1e-3)
    ts = cfg["train_setup"]
    ts.update(train_subset_ratio=None, eval_every=2_000_000, reset_every=50_000_000)
    ts["train_config"].update(num_envs=8192, num_timesteps=50_000_000, discounting=0.95, entropy_cost=5e-2)
    return cfg


NAMED_CONFIGS = {"rodent-full-clips": default_config, "rodent-sps-per-actor": sps_per_actor_config}


def named_config(name: str) -> dict[str, Any]:
    """`--config-name` of the reference's hydra CLI (train.py:56): one of the shipped rodent configurations by file stem."""
    try:
        return NAMED_CONFIGS[name]()
    except KeyError:
        raise KeyError(f"unknown configuration {name!r}: one of {sorted(NAMED_CONFIGS)}") from None


def _deep_update(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(path: str | None = None, overrides: list[str] | None = None, name: str | None = None) -> dict[str, Any]:
    cfg = named_config(name) if name else default_config()
    if path:
        with open(path) as f:
            _deep_update(cfg, yaml.safe_load(f) or {})
    for ov in overrides or []:
        key, _, val = ov.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def reward_vector(rw: dict[str, Any]) -> np.ndarray:
    d = dict(var_coeff=5e-2, jerk_coeff=5e-4)     # RewardConfig's defaults (reward.py:52-53) for reward_weights dicts without the keys
    d.update(rw)
    d["healthy_z_lo"], d["healthy_z_hi"] = d["healthy_z_range"]
    d["pen0"], d["pen1"], d["pen2"] = d["penalty_pos_distance_scale"]
    return np.array([float(d[k]) for k in REWARD_F], dtype=np.float64)


def clone(cfg: dict) -> dict:
    return copy.deepcopy(cfg)
