"""Evaluation roll-outs — mirror of the evaluator the reference's train loop uses.

Reference: track_mjx/agent/mlp_ppo/ppo.py:83-124 (run_evaluation, monkey-patched onto brax `acting.Evaluator`),
:629-668 (construction: eval env wrapped like the training env, `num_eval_envs`, `deterministic_eval`), :744-758
(called once per epoch on process 0).  brax 0.12.3 is not vendored; restated from its published definition:
`acting.Evaluator` resets `num_eval_envs` envs, unrolls the policy for `episode_length // action_repeat` steps through
`envs.training.EvalWrapper`, which keeps per env
    episode_metrics[name] += metric[name] * active_episodes        (metrics = the env's 20 metrics + "reward")
    episode_steps         = where(active_episodes, info["steps"], episode_steps)
    active_episodes      *= 1 - done
i.e. sums over the FIRST episode of every env only (the auto-reset wrapper keeps stepping afterwards).
The roll-out itself is the product path: tmjx_reset / tmjx_step of the eval env + the policy GEMMs.
"""
from __future__ import annotations

import time
from typing import Callable

import torch


class EvalWrapper:
    """brax.envs.wrappers.training.EvalWrapper over a wrapped MultiClipTracking (batched, stateful)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, rng, *a, **kw):
        st = self.env.reset(rng, *a, **kw)
        n, dev = self.env.num_envs, self.env.device
        names = list(st.metrics.keys()) + ["reward"]
        self.episode_metrics = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in names}
        self.active_episodes = torch.ones(n, dtype=torch.float32, device=dev)
        self.episode_steps = torch.zeros(n, dtype=torch.float32, device=dev)
        return st

    def step(self, state, action):
        nstate = self.env.step(state, action)
        active = self.active_episodes
        steps = nstate.info["steps"].float()
        self.episode_steps = torch.where(active > 0, steps, self.episode_steps)
        for k in self.episode_metrics:
            v = nstate.reward if k == "reward" else nstate.metrics[k]
            self.episode_metrics[k] += v * active
        self.active_episodes = active * (1.0 - nstate.done)
        return nstate


class Evaluator:
    """acting.Evaluator(eval_env, make_policy(deterministic=...), num_eval_envs, episode_length, action_repeat, key)."""

    def __init__(self, eval_env, policy: Callable, *, episode_length: int, action_repeat: int = 1, seed: int = 0):
        self.env = EvalWrapper(eval_env)
        self.policy = policy
        self.steps_per_unroll = int(episode_length) * eval_env.num_envs
        self.unroll_length = int(episode_length) // int(action_repeat)
        self.gen = torch.Generator().manual_seed(seed)
        self.eval_walltime = 0.0

    @torch.no_grad()
    def run_evaluation(self, training_metrics: dict | None = None, aggregate_episodes: bool = True, data_split: str = "") -> dict:
        t0 = time.time()
        st = self.env.reset(self.gen)
        for _ in range(self.unroll_length):
            action, _ = self.policy(st.obs)
            st = self.env.step(st, action)
        torch.cuda.synchronize(self.env.device) if self.env.device.type == "cuda" else None
        dt = time.time() - t0
        prefix = f"{data_split}/" if data_split else ""
        metrics = {}
        for suffix, fn in (("", torch.mean), ("_std", lambda x: torch.std(x, unbiased=False))):
            for name, value in self.env.episode_metrics.items():
                metrics[f"eval/{prefix}episode_{name}{suffix}"] = float(fn(value)) if aggregate_episodes else value.cpu().numpy()
        metrics[f"eval/{prefix}avg_episode_length"] = float(self.env.episode_steps.mean())
        metrics[f"eval/{prefix}epoch_eval_time"] = dt
        metrics[f"eval/{prefix}sps"] = self.steps_per_unroll / dt
        self.eval_walltime += dt
        return {f"eval/{prefix}walltime": self.eval_walltime, **(training_metrics or {}), **metrics}
