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


def default_walker(config: str = "rodent-full-clips"):
    """Walker + one of the reference's two shipped rodent configurations (track_mjx_amd/config.py: NAMED_CONFIGS)."""
    cfg = _config.named_config(config)
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
        clip_length=rc["clip_length"], traj_length=rc["traj_length"], window=rw.get("var_window_size", 50),
        episode_length=episode_length, reward_f=_config.reward_vector(rw), auto_reset=auto_reset)


def make_oracle(blob, clip=None, precision="f32"):
    from oracle.oracle import Oracle
    O = Oracle(blob, precision)
    if clip is not None:
        O.set_clips(clip.as_dict())
    return O


def make_env_and_oracle(num_envs=64, n_clips=4, device="cuda:0", wrappers=True, seed=0, precision="f32", episode_length=195,
                        config="rodent-full-clips"):
    from track_mjx_amd.environment import MultiClipTracking, RewardConfig, wrap
    w, cfg = default_walker(config)
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


PHYS_ROWS = ("qpos", "qvel", "act", "qacc_warmstart", "time")


def spread_sample(n: int, k: int) -> list[int]:
    """k env indices of a launch of n envs: both ends of the launch (first / last three: where a grid-tail or record-stride bug would show) and
    an even spread in between."""
    k = min(k, n)
    edge = [0, 1, 2, n - 3, n - 2, n - 1]
    mid = [int(round(x)) for x in np.linspace(3, n - 4, max(k - len(edge), 0))]
    return sorted(set(i for i in edge + mid if 0 <= i < n))


def oracle_substep_sample(envs, O32, O64, rng, scale=0.3, per_group=22, substeps=3):
    """Teacher-forced comparison at FULL launch size: every env group takes `substeps` single-substep launches of ALL its envs (tmjx_physics);
    before each, the current device state of a SAMPLE of envs (spread_sample: first / last envs of every group + a spread) is copied into the
    float64 and float32 oracles, which take the same substep.  Returns the per-(env, substep) relative errors against the float64 oracle:
    {"qpos" | "qvel": (hip, f32_oracle)}.  The envs' states advance by `substeps` substeps (call it last)."""
    import torch
    out = {k: ([], []) for k in ("qpos", "qvel")}
    for env in envs:
        n = env.num_envs
        idx = spread_sample(n, per_group)
        ti = torch.as_tensor(idx, device=env.device)
        for _ in range(substeps):
            st = {k: env.rows(k)[:, ti].cpu().numpy().astype(np.float64) for k in PHYS_ROWS}
            ok = [j for j in range(len(idx)) if all(np.isfinite(st[k][:, j]).all() for k in PHYS_ROWS)]
            a = np.clip(rng.normal(size=(38, n)) * scale, -1, 1).astype(np.float32)
            env.physics(torch.from_numpy(a).to(env.device), 1)
            torch.cuda.synchronize()
            got = {k: env.rows(k)[:, ti].cpu().numpy() for k in ("qpos", "qvel")}
            for j in ok:
                res = {}
                for O, tag in ((O64, "f64"), (O32, "f32")):
                    d = O.new_data(st["qpos"][:, j], st["qvel"][:, j])
                    for k in PHYS_ROWS:
                        O.set(d, k, st[k][:, j])
                    O.step(d, a[:, idx[j]].astype(np.float64))
                    res[tag] = {k: O.get(d, k) for k in ("qpos", "qvel")}
                if not all(np.isfinite(res["f64"][k]).all() for k in res["f64"]):
                    continue
                for k in ("qpos", "qvel"):
                    out[k][0].append(rel_err(got[k][:, j], res["f64"][k])); out[k][1].append(rel_err(res["f32"][k], res["f64"][k]))
    return {k: (np.array(v[0]), np.array(v[1])) for k, v in out.items()}


def assert_substep_sample_bounds(errs, min_samples=100):
    """Bounds of tests/test_gpu_parity_strict.py, widened for the small sample (a few hundred env-substeps instead of 2560)."""
    for k, (g, f) in errs.items():
        fin = np.isfinite(f)
        assert len(g) >= min_samples and np.isfinite(g[fin]).all(), (k, len(g))
        g, f = g[fin], f[fin]
        assert np.median(g) <= 1e-5, (k, np.median(g))
        assert np.median(g) <= 2 * np.median(f) + 1e-7, (k, np.median(g), np.median(f))
        assert np.quantile(g, 0.9) <= 3 * np.quantile(f, 0.9) + 1e-5, (k, np.quantile(g, 0.9), np.quantile(f, 0.9))
        assert g.max() <= 4 * f.max() + 1e-4, (k, g.max(), f.max())


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
