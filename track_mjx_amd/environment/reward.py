"""compute_tracking_rewards mirror (reference: track_mjx/environment/task/reward.py:359-485).

In the reference this is a pure function of one env's `mjx.Data`, the gathered reference frame, the action and `info`, vmapped by
brax.  Here the 18 terms are what the K3 kernel computes for every env of the batch (csrc/env_core.h, C-ABI `tmjx_reward_obs`);
this entry runs that kernel on a COPY of the env's state so that, like the reference function, it has no side effects: the ring
buffer, step counter, done flag and auto-reset snapshot of the env are left alone.  A `reference_frame` passed by the caller (the
reference's call, single_clip_tracking.py:239-246) is what the terms are computed from (`tmjx_reward_frame`).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import hip as _hip

TERMS = ("pos_reward", "quat_reward", "joint_reward", "angvel_reward", "bodypos_reward", "endeff_reward", "ctrl_cost", "ctrl_diff_cost",
         "energy_cost", "too_far", "bad_pose", "bad_quat", "fall", "joint_distance", "summed_pos_distance", "quat_distance",
         "action_variance_cost", "jerk_cost")
# rows of the kernel's metrics buffer (environment/task.py METRIC_NAMES) holding the terms; the step metrics store the five costs
# NEGATED (single_clip_tracking.py:295-316), the reward function returns them positive
_ROW = {"pos_reward": 0, "quat_reward": 1, "joint_reward": 2, "angvel_reward": 3, "bodypos_reward": 4, "endeff_reward": 5, "ctrl_cost": 6,
        "ctrl_diff_cost": 7, "energy_cost": 8, "too_far": 10, "bad_pose": 11, "bad_quat": 12, "fall": 13, "joint_distance": 15,
        "summed_pos_distance": 16, "quat_distance": 17, "action_variance_cost": 18, "jerk_cost": 19}
_NEGATED = {"ctrl_cost", "ctrl_diff_cost", "energy_cost", "action_variance_cost", "jerk_cost"}


_FRAME_LEAVES = ("position", "quaternion", "joints", "body_positions", "angular_velocity")


def _frame_leaves(reference_frame, env):
    """The five leaves of the caller's gathered frame as contiguous fp32 device tensors [n][...] (ReferenceClip of ONE frame per env, as
    `info["reference_frame"]` holds it after multi_clip_tracking.py:98-109 / single_clip_tracking.py:223-225 under brax's vmap)."""
    n, L = env.num_envs, env.layout
    want = {"position": (n, 3), "quaternion": (n, 4), "joints": (n, L.nq - 7), "body_positions": (n, L.nbody - 1, 3), "angular_velocity": (n, 3)}
    out = []
    for k in _FRAME_LEAVES:
        v = reference_frame[k] if isinstance(reference_frame, dict) else getattr(reference_frame, k)
        v = torch.as_tensor(v, dtype=torch.float32).to(env.device)
        if n == 1 and v.dim() == len(want[k]) - 1:
            v = v.unsqueeze(0)            # a single env's un-batched frame
        if tuple(v.shape) != want[k]:
            raise ValueError(f"reference_frame.{k} must have shape {want[k]}, got {tuple(v.shape)}")
        out.append(v.contiguous())
    return out


def compute_tracking_rewards(data, reference_frame, walker, action: torch.Tensor, info: dict | None, reward_config=None):
    """-> the reference's 18-tuple (TERMS order), each a [num_envs] tensor.

    data: `State.pipeline_state` of a MultiClipTracking env (views of its device buffers; carries the env).  `reference_frame`: the
    gathered frame a reference caller passes (`ReferenceClip`-like object or dict with the leaves position [n,3], quaternion [n,4], joints
    [n,67], body_positions [n,67,3], angular_velocity [n,3]; single_clip_tracking.py:239-246) — the rewards are then computed FROM IT
    (C-ABI `tmjx_reward_frame`); None: the kernel gathers frame `_get_cur_frame()` of clip `info["clip_idx"]` from the env's resident
    clip table itself (multi_clip_tracking.py:98-109, single_clip_tracking.py:223-225).  `action` [n, nu] or [nu][n].  `info` = the env's info BEFORE this
    step's bookkeeping: like SingleClipTracking.step (single_clip_tracking.py:227-234) the kernel first sets prev_ctrl = action (so
    ctrl_diff_cost is 0, the reference's own quirk) and writes the action into the ring buffer, on the copy."""
    env = data["_env"]() if isinstance(data, dict) else data
    if env is None:
        raise ValueError("the env of this pipeline_state no longer exists")
    if walker is not None and walker is not env.walker:
        raise ValueError("walker differs from the env's")
    if reward_config is not None and reward_config is not env._reward_config and reward_config != env._reward_config:
        raise ValueError("reward_config differs from the one compiled into the env's handle")
    n, L = env.num_envs, env.layout
    a = action.t() if (action.shape == (n, L.nu) and not (n == L.nu and action.stride(0) == 1)) else action
    a = a.to(device=env.device, dtype=torch.float32).contiguous()
    if a.shape != (L.nu, n):
        raise ValueError(f"action must be [{n},{L.nu}] or [{L.nu},{n}]")
    f32 = dict(dtype=torch.float32, device=env.device)
    st, ist, ws = env.state_buf.clone(), env.istate_buf.clone(), torch.empty_like(env.workspace)
    obs, met = torch.empty_like(env.obs_buf), torch.empty((L.n_metrics, n), **f32)
    rew, done, trunc = torch.empty(n, **f32), torch.empty(n, **f32), torch.empty(n, **f32)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(env.device):
        stream = C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)
        if reference_frame is None:
            _hip.check(env._L.tmjx_reward_obs(env._handle, p(st), p(ist), p(a), p(obs), p(rew), p(done), p(trunc), p(met), p(ws), n, stream), "tmjx_reward_obs")
        else:
            fr = _frame_leaves(reference_frame, env)
            _hip.check(env._L.tmjx_reward_frame(env._handle, p(st), p(ist), p(a), *[p(v) for v in fr], p(obs), p(rew), p(done), p(trunc), p(met), n, stream),
                       "tmjx_reward_frame")
    return tuple(-met[_ROW[k]] if k in _NEGATED else met[_ROW[k]] for k in TERMS)
