"""Shared helpers for the tests (and __graft_entry__.smoke / bench.py's cpu_baseline leg)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from track_mjx_amd import clips as _clips  # noqa: E402
from track_mjx_amd import config as _config  # noqa: E402
from track_mjx_amd import walker as _walker  # noqa: E402


def default_walker():
    cfg = _config.default_config()
    return _walker.Rodent(**cfg["walker_config"]), cfg


def default_blob(w=None, cfg=None, *, episode_length=195, auto_reset=True, n_frames=None, iterations=None, timestep=None):
    if w is None:
        w, cfg = default_walker()
    ea = cfg["env_config"]["env_args"]
    rw = cfg["env_config"]["reward_weights"]
    rc = cfg["reference_config"]
    return _walker.build_blob(
        w, n_frames=n_frames or ea["physics_steps_per_control_step"], iterations=iterations or ea["iterations"],
        ls_iterations=iterations or ea["ls_iterations"], timestep=timestep or ea["mj_model_timestep"], mocap_hz=ea["mocap_hz"],
        clip_length=rc["clip_length"], traj_length=rc["traj_length"], window=rw["var_window_size"],
        episode_length=episode_length, reward_f=_config.reward_vector(rw), auto_reset=auto_reset)


def make_oracle(blob, clip=None, precision="f32"):
    from oracle.oracle import Oracle
    O = Oracle(blob, precision)
    if clip is not None:
        O.set_clips(clip.as_dict())
    return O


def make_env_and_oracle(num_envs=64, n_clips=4, device="cuda:0", wrappers=True, seed=0, precision="f32", episode_length=195):
    from track_mjx_amd.environment import MultiClipTracking, RewardConfig, wrap
    w, cfg = default_walker()
    cl = _clips.make_synthetic_clips(w.model, n_clips, seed=seed)
    ea = cfg["env_config"]["env_args"]
    env = MultiClipTracking(cl, w, RewardConfig(**cfg["env_config"]["reward_weights"]), **ea, **cfg["reference_config"],
                            num_envs=num_envs, device=device)
    if wrappers:
        env = wrap(env, episode_length=episode_length)
    O = make_oracle(env._blob, cl, precision)
    return env, O, cl


def rel_err(a, b, axis=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max(axis) / (np.abs(b).max(axis) + 1e-12)


class StubEnv:
    """Just enough of MultiClipTracking for PPOLearner on the CPU (tests of the learner's host logic and collectives): sizes and a
    device; the roll-out buffers are filled by the test, no env is ever stepped."""

    class _Layout:
        ref_obs_size = 470

    def __init__(self, num_envs: int, obs: int = 696, ref: int = 470, nu: int = 38):
        import torch
        self.num_envs, self.device = num_envs, torch.device("cpu")
        self.observation_size, self.action_size = obs, nu
        self.layout = StubEnv._Layout()
        self.layout.ref_obs_size = ref


def torch_gae(truncation, termination, rewards, values, bootstrap_value, lambda_=1.0, discount=0.99):
    """compute_gae (track_mjx/agent/mlp_ppo/losses.py:39-100) in torch, for CPU tests of the learner (the product runs tmjx_gae)."""
    import torch
    T = rewards.shape[0]
    tm = 1 - truncation
    v1 = torch.cat([values[1:], bootstrap_value[None]], 0)
    deltas = (rewards + discount * (1 - termination) * v1 - values) * tm
    acc = torch.zeros_like(bootstrap_value)
    out = []
    for t in range(T - 1, -1, -1):
        acc = deltas[t] + discount * (1 - termination[t]) * tm[t] * lambda_ * acc
        out.append(acc)
    vs = torch.stack(out[::-1], 0) + values
    vs1 = torch.cat([vs[1:], bootstrap_value[None]], 0)
    adv = (rewards + discount * (1 - termination) * vs1 - values) * tm
    return vs.detach(), adv.detach()
