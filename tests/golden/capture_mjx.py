#!/usr/bin/env python3
"""Capture MJX golden vectors for BASELINE config 1 — BUILD CONTAINER ONLY, never shipped to or run on the GPU box.

STATUS: cannot run today.  jax / mujoco / mujoco-mjx / brax are not installed in this image (re-probed every session:
`python -c "import jax, mujoco, brax"` -> ModuleNotFoundError) and there is no network, so the oracle is "parity unpinned" at the
physics boundary.  This script exists so that the moment those imports work, ONE command pins it:

    python tests/golden/capture_mjx.py            # writes tests/golden/mjx_golden_cfg1.npz (commit it)

What it does (last paragraph): imports the REFERENCE env (track_mjx.environment.task.multi_clip_tracking
from /root/reference, where it lies — nothing is copied), builds it on the same synthetic clip table the tests use
(track_mjx_amd.clips.make_synthetic_clips, seed 0, 1 clip), wraps it like the learner does (wrappers.wrap, episode_length 195),
resets from jax key 0 (clip index, start frame and noise are the env's own draws, logged), steps 195 control steps with logged N(0,1) actions
(clipped to the control range by the model) and dumps per control step: qpos, qvel, act, qacc_warmstart, time, xpos, torso xmat,
qfrc_actuator, obs, reward, done, truncation, the 20 metrics, contact.dist, contact.geom, the non-zero pattern of efc_J,
efc_force, solver_niter.  A second block steps `mjx.step` substep by substep (teacher-forced from the logged states) so that the
per-substep comparison of tests/test_gpu_parity_strict.py can be made against MJX itself.
tests/test_mjx_golden.py picks the file up automatically (it is skipped while the file does not exist).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
REFERENCE = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "mjx_golden_cfg1.npz"


def main() -> int:
    try:
        import jax
        import jax.numpy as jp
        import mujoco  # noqa: F401
        from mujoco import mjx
        import brax  # noqa: F401
    except ImportError as e:
        print(f"capture_mjx: {e}; the MJX stack is not installed here — nothing captured (parity stays unpinned)")
        return 2
    if not REFERENCE.exists():
        print("capture_mjx: /root/reference is not present on this machine")
        return 2
    jax.config.update("jax_platform_name", "cpu")
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(REFERENCE))
    from track_mjx.environment import wrappers as ref_wrappers                      # the reference itself, imported where it lies
    from track_mjx.environment.task.multi_clip_tracking import MultiClipTracking
    from track_mjx.environment.task.reward import RewardConfig
    from track_mjx.environment.walker.rodent import Rodent
    from track_mjx.io.load import ReferenceClip

    from track_mjx_amd import clips as _clips, config as _config
    from track_mjx_amd.walker import Rodent as OurRodent

    cfg = _config.default_config()
    ours = OurRodent(**cfg["walker_config"])
    table = _clips.make_synthetic_clips(ours.model, 1, seed=0)
    d = table.as_dict()
    zeros = lambda *shape: jp.zeros(shape, dtype=jp.float32)  # noqa: E731
    C, F = d["position"].shape[:2]
    clip = ReferenceClip(position=jp.asarray(d["position"]), quaternion=jp.asarray(d["quaternion"]), joints=jp.asarray(d["joints"]),
                         body_positions=jp.asarray(d["body_positions"]), velocity=zeros(C, F, 3),
                         angular_velocity=jp.asarray(d["angular_velocity"]), joints_velocity=zeros(C, F, d["joints"].shape[-1]),
                         body_quaternions=zeros(C, F, d["body_positions"].shape[2], 4))
    walker = Rodent(**cfg["walker_config"])
    env = MultiClipTracking(reference_clip=clip, walker=walker, reward_config=RewardConfig(**cfg["env_config"]["reward_weights"]),
                            **cfg["env_config"]["env_args"], **cfg["reference_config"])
    wenv = ref_wrappers.wrap(env, episode_length=195, action_repeat=1)

    # compile-time constants the model compiler restates: pin them too
    sys_ = env.sys
    model_consts = {f"model_{k}": np.asarray(getattr(sys_, k)) for k in
                    ("body_mass", "body_inertia", "body_ipos", "body_iquat", "body_pos", "body_quat", "dof_damping", "dof_armature",
                     "dof_invweight0", "body_invweight0", "jnt_range", "jnt_stiffness", "qpos0", "qpos_spring", "actuator_gainprm",
                     "actuator_gear", "geom_size", "geom_pos", "geom_quat", "geom_friction")}
    model_consts["model_meaninertia"] = np.asarray(sys_.stat.meaninertia)

    # the wrapped (vmapped, 1 env) reset as the learner calls it; the clip index / start frame it drew are logged below, the noise
    # follows from the key (track_mjx_amd/jax_random.py reproduces the draws: reset_draws_batch)
    rng = jax.random.PRNGKey(0)
    state = jax.jit(wenv.reset)(rng[None])
    step = jax.jit(wenv.step)
    acts = np.clip(np.random.default_rng(0).normal(size=(195, 38)), -1, 1).astype(np.float32)
    log: dict[str, list] = {k: [] for k in ("qpos", "qvel", "act", "qacc_warmstart", "time", "xpos", "xmat_torso", "qfrc_actuator", "obs",
                                              "reward", "done", "truncation", "metrics", "contact_dist", "contact_geom", "efc_J_nnz",
                                              "efc_force", "solver_niter", "clip_idx", "start_frame")}
    metric_names = None
    torso = int(walker._torso_idx) if hasattr(walker, "_torso_idx") else 3
    for t in range(195):
        state = step(state, jp.asarray(acts[t])[None])
        ps = state.pipeline_state
        for k in ("qpos", "qvel", "act", "qacc_warmstart", "time", "xpos", "qfrc_actuator", "efc_force"):
            log[k].append(np.asarray(getattr(ps, k))[0])
        log["xmat_torso"].append(np.asarray(ps.xmat)[0, torso].reshape(9))
        log["obs"].append(np.asarray(state.obs)[0]); log["reward"].append(float(state.reward[0])); log["done"].append(float(state.done[0]))
        log["truncation"].append(float(state.info["truncation"][0]))
        metric_names = metric_names or sorted(state.metrics)
        log["metrics"].append(np.array([float(state.metrics[k][0]) for k in metric_names]))
        log["contact_dist"].append(np.asarray(ps.contact.dist)[0]); log["contact_geom"].append(np.asarray(ps.contact.geom)[0])
        log["efc_J_nnz"].append((np.asarray(ps.efc_J)[0] != 0).astype(np.uint8))
        log["solver_niter"].append(int(np.asarray(ps.solver_niter)[0]))
        log["clip_idx"].append(int(state.info["clip_idx"][0])); log["start_frame"].append(int(state.info["start_frame"][0]))

    # teacher-forced substeps: mjx.step from every logged pre-step state with the logged action, 10 substeps, all intermediate states
    mstep = jax.jit(mjx.step)
    sub = {k: [] for k in ("qpos", "qvel", "act", "qacc_warmstart", "solver_niter")}
    data0 = mjx.make_data(sys_)
    prev = None
    for t in range(195):
        if prev is None:
            prev = {k: log[k][t] for k in ("qpos", "qvel", "act", "qacc_warmstart", "time")}
            continue
        dd = data0.replace(qpos=jp.asarray(prev["qpos"]), qvel=jp.asarray(prev["qvel"]), act=jp.asarray(prev["act"]),
                           qacc_warmstart=jp.asarray(prev["qacc_warmstart"]), time=jp.asarray(prev["time"]), ctrl=jp.asarray(acts[t]))
        for _ in range(10):
            dd = mstep(sys_, dd)
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                sub[k].append(np.asarray(getattr(dd, k)))
            sub["solver_niter"].append(int(dd.solver_niter))
        prev = {k: log[k][t] for k in ("qpos", "qvel", "act", "qacc_warmstart", "time")}
    out = {k: np.asarray(v) for k, v in log.items()}
    out.update({f"sub_{k}": np.asarray(v) for k, v in sub.items()})
    out.update(model_consts)
    out["actions"] = acts
    out["reset_key"] = np.asarray(rng)
    out["metric_names"] = np.array(metric_names)
    out["versions"] = np.array([f"jax {jax.__version__}", f"mujoco {mujoco.__version__}", f"brax {brax.__version__}"])
    np.savez_compressed(OUT, **out)
    print(f"capture_mjx: wrote {OUT} ({OUT.stat().st_size / 1e6:.2f} MB)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
