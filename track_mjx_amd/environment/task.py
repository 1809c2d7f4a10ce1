"""MultiClipTracking: the reference's task env interface on top of libtmjx_hip.so.

Mirrors (paths relative to /root/reference/track_mjx):
  environment/task/multi_clip_tracking.py:13-109   constructor signature, reset(rng, clip_idx)
  environment/task/single_clip_tracking.py:207-320 step(state, action)
  environment/task/reward.py:15-54                 RewardConfig

Differences that are inherent to the platform, not to the maths:
  * the env is *batched and stateful*: all `num_envs` envs live in device buffers laid out
    [field][env] (include/tmjx.h); `State` holds views of those buffers and `step` updates them in
    place (brax's functional State/vmap machinery is what this replaces);
  * JAX threefry keys are replaced by a `torch.Generator` (or explicit clip_idx/start_frame/noise
    tensors) — same-seed parity with JAX is out of scope;
  * PyTorch is used only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch

from .. import config as _config
from .. import hip as _hip
from ..clips import ReferenceClip
from ..walker import Rodent, build_blob

METRIC_NAMES = ("pos_reward", "quat_reward", "joint_reward", "angvel_reward", "bodypos_reward", "endeff_reward",
                "ctrl_cost", "ctrl_diff_cost", "energy_cost", "done", "too_far", "bad_pose", "bad_quat", "fall", "nan",
                "joint_distance", "summed_pos_distance", "quat_distance", "var_cost", "jerk_cost")


@dataclass
class RewardConfig:
    """Weights and scales of the imitation reward (reference: task/reward.py:15-54)."""
    too_far_dist: float
    bad_pose_dist: float
    bad_quat_dist: float
    ctrl_cost_weight: float
    ctrl_diff_cost_weight: float
    energy_cost_weight: float
    pos_reward_weight: float
    quat_reward_weight: float
    joint_reward_weight: float
    angvel_reward_weight: float
    bodypos_reward_weight: float
    endeff_reward_weight: float
    healthy_z_range: tuple
    pos_reward_exp_scale: float
    quat_reward_exp_scale: float
    joint_reward_exp_scale: float
    angvel_reward_exp_scale: float
    bodypos_reward_exp_scale: float
    endeff_reward_exp_scale: float
    penalty_pos_distance_scale: Any
    var_window_size: int = 50
    var_coeff: float = 5e-2
    jerk_coeff: float = 5e-4

    def vector(self) -> np.ndarray:
        return _config.reward_vector(self.__dict__)


@dataclass
class State:
    """Batched env state. Tensors are views of the device buffers the HIP kernels update in place."""
    pipeline_state: dict          # qpos [n,nq], qvel [n,nv], act, qacc_warmstart, time, xpos [n,nbody,3] (views)
    obs: torch.Tensor             # [n, obs_size] (transposed view of the [obs][env] buffer)
    reward: torch.Tensor          # [n]
    done: torch.Tensor            # [n]
    metrics: dict                 # name -> [n]
    info: dict = field(default_factory=dict)


def _ptr(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class MultiClipTracking:
    """Batched multi-clip tracking env (reference: task/multi_clip_tracking.py:13)."""

    def __init__(self, reference_clip: ReferenceClip | None, walker: Rodent, reward_config: RewardConfig,
                 physics_steps_per_control_step: int, reset_noise_scale: float, solver: str = "cg", iterations: int = 4,
                 ls_iterations: int = 4, mj_model_timestep: float = 0.002, mocap_hz: int = 50, clip_length: int = 250,
                 random_init_range: int = 50, traj_length: int = 5, *, num_envs: int = 1, device: str | torch.device = "cuda",
                 episode_length: int | None = None, auto_reset: bool = False, share_clips_with: "MultiClipTracking | None" = None, **kwargs: Any):
        if solver.lower() != "cg":
            raise NotImplementedError("only the CG solver is built (every shipped reference config uses solver: cg)")
        self.walker = walker
        self._reward_config = reward_config
        self._n_frames = int(physics_steps_per_control_step)
        self._reset_noise_scale = float(reset_noise_scale)
        self._mocap_hz, self._clip_length, self._ref_len = int(mocap_hz), int(clip_length), int(traj_length)
        self._random_init_range = int(random_init_range)
        self._opts = dict(iterations=int(iterations), ls_iterations=int(ls_iterations), timestep=float(mj_model_timestep))
        self._steps_for_cur_frame = (1.0 / (mocap_hz * mj_model_timestep)) / physics_steps_per_control_step
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self._reference_clips = reference_clip
        self._n_clips = int(reference_clip.position.shape[0]) if reference_clip is not None else 0
        self._L = _hip.lib()  # raises loudly if the HIP extension is missing
        self._handle = C.c_void_p()
        self._physics_events = None   # list => step() records (start, end) HIP events around the physics kernel
        self._episode_length = int(episode_length) if episode_length is not None else (1 << 30)
        self._auto_reset = bool(auto_reset)
        # the env groups of one rank read ONE resident clip table (tmjx_clips_share): this env uses `share_clips_with`'s device arrays
        self._clip_owner = share_clips_with
        self._create_handle()
        self._alloc()

    # ---- handle / buffers
    def _create_handle(self) -> None:
        if self._handle:
            self._L.tmjx_model_destroy(self._handle)
            self._handle = C.c_void_p()
        blob = build_blob(self.walker, n_frames=self._n_frames, mocap_hz=self._mocap_hz, clip_length=self._clip_length,
                          traj_length=self._ref_len, window=int(self._reward_config.var_window_size),
                          episode_length=self._episode_length, reward_f=self._reward_config.vector(),
                          auto_reset=self._auto_reset, **self._opts)
        self._blob = blob
        with torch.cuda.device(self.device):
            _hip.check(self._L.tmjx_model_create(blob, len(blob), C.byref(self._handle)), "tmjx_model_create")
            self.layout = _hip.Layout()
            _hip.check(self._L.tmjx_layout(self._handle, C.byref(self.layout)), "tmjx_layout")
            if self._clip_owner is not None:
                _hip.check(self._L.tmjx_clips_share(self._handle, self._clip_owner._handle), "tmjx_clips_share")
            elif self._reference_clips is not None:
                c = self._reference_clips
                arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in
                        (c.position, c.quaternion, c.joints, c.body_positions, c.angular_velocity)]
                if arrs[3].shape[2] != self.walker.nbody - 1:
                    raise ValueError("clip body_positions must have nbody-1 rows (reference aligns them with xpos[1:])")
                n_clips, n_frames = arrs[0].shape[:2]
                _hip.check(self._L.tmjx_clips_upload(self._handle, *[a.ctypes.data_as(C.c_void_p) for a in arrs], n_clips, n_frames),
                           "tmjx_clips_upload")

    def _alloc(self) -> None:
        n, L, dev = self.num_envs, self.layout, self.device
        self.state_buf = torch.zeros((L.state_rows, n), dtype=torch.float32, device=dev)
        self.istate_buf = torch.zeros((L.istate_rows, n), dtype=torch.int32, device=dev)
        self.workspace = torch.zeros((L.ws_rows, n), dtype=torch.float32, device=dev)
        self.obs_buf = torch.zeros((L.obs_size, n), dtype=torch.float32, device=dev)
        self.reward_buf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.done_buf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.trunc_buf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.metrics_buf = torch.zeros((L.n_metrics, n), dtype=torch.float32, device=dev)

    def __del__(self):
        try:
            if self._handle:
                self._L.tmjx_model_destroy(self._handle)
        except Exception:
            pass

    # ---- brax Env protocol attributes used by callers (ppo.py:482-512, train.py:221-225)
    @property
    def observation_size(self) -> int:
        return int(self.layout.obs_size)

    @property
    def action_size(self) -> int:
        return int(self.layout.nu)

    @property
    def dt(self) -> float:
        return self._opts["timestep"] * self._n_frames

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _state(self) -> State:
        L, sb = self.layout, self.state_buf
        ps = {
            "qpos": sb[L.qpos:L.qpos + L.nq].t(), "qvel": sb[L.qvel:L.qvel + L.nv].t(), "act": sb[L.act:L.act + L.nu].t(),
            "qacc_warmstart": sb[L.qacc_warmstart:L.qacc_warmstart + L.nv].t(), "time": sb[L.time],
            "xpos": sb[L.xpos:L.xpos + 3 * L.nbody].t().reshape(self.num_envs, L.nbody, 3),
            "qfrc_actuator": sb[L.qfrc_actuator:L.qfrc_actuator + L.nv].t(),
            "_env": weakref.ref(self),        # environment/reward.py: compute_tracking_rewards(data, ...) finds the handle through it
        }
        info = {
            "truncation": self.trunc_buf, "steps": sb[L.steps_f], "clip_idx": self.istate_buf[L.i_clip_idx],
            "start_frame": self.istate_buf[L.i_start_frame], "buffer_index": self.istate_buf[L.i_buffer_index],
            "prev_ctrl": sb[L.prev_ctrl:L.prev_ctrl + L.nu].t(),
            "reference_obs_size": int(L.ref_obs_size), "proprioceptive_obs_size": int(L.obs_size - L.ref_obs_size),
            "first_obs": sb[L.first_obs:L.first_obs + L.obs_size].t(),
        }
        metrics = {name: self.metrics_buf[i] for i, name in enumerate(METRIC_NAMES)}
        return State(ps, self.obs_buf.t(), self.reward_buf, self.done_buf, metrics, info)

    # ---- reset / step
    def reset(self, rng=None, clip_idx: torch.Tensor | None = None, *,
              start_frame: torch.Tensor | None = None, qpos_noise: torch.Tensor | None = None,
              qvel_noise: torch.Tensor | None = None) -> State:
        """reset(rng, clip_idx=None) (reference: task/multi_clip_tracking.py:74-96).

        start_frame ~ randint[0,44) and clip_idx ~ randint[0,n_clips) as in the reference; the noise is
        U(-reset_noise_scale, +reset_noise_scale) for qpos[nq] and qvel[nv].  Explicit tensors override the draws
        (layout: clip_idx/start_frame [n] int32, qpos_noise [nq][n], qvel_noise [nv][n]).  `rng`: a torch.Generator / int seed
        (torch's generator), or a jax PRNG key as a uint32 numpy array ([2] or [n, 2]): then clip, start frame and noise are the
        reference's own draws from that key (threefry, jax_random.py)."""
        n, L, dev = self.num_envs, self.layout, self.device
        import numpy as _np
        if isinstance(rng, _np.ndarray) and rng.dtype == _np.uint32 and rng.shape[-1] == 2:
            # a jax PRNG key ([2] uint32: split into one key per env as brax's vmapped reset receives them, or [n, 2] per-env
            # keys): the draws of the reference from the same key(s) (jax_random.py)
            from .. import jax_random as _jr
            keys = _jr.split(rng, n) if rng.ndim == 1 else rng
            if keys.shape != (n, 2):
                raise ValueError(f"per-env keys must be [{n}, 2] uint32")
            ci_, sf_, qn_, vn_ = _jr.reset_draws_batch(keys, max(self._n_clips, 1), int(L.nq), int(L.nv), self._reset_noise_scale)
            clip_idx = torch.from_numpy(ci_) if clip_idx is None else clip_idx
            start_frame = torch.from_numpy(sf_) if start_frame is None else start_frame
            qpos_noise = torch.from_numpy(qn_) if qpos_noise is None else qpos_noise
            qvel_noise = torch.from_numpy(vn_) if qvel_noise is None else qvel_noise
            rng = None
        g = rng if isinstance(rng, torch.Generator) else torch.Generator(device="cpu").manual_seed(int(rng or 0))
        if start_frame is None:
            start_frame = torch.randint(0, 44, (n,), generator=g, dtype=torch.int32)
        if clip_idx is None:
            clip_idx = torch.randint(0, max(self._n_clips, 1), (n,), generator=g, dtype=torch.int32)
        s = self._reset_noise_scale
        if qpos_noise is None:
            qpos_noise = (torch.rand((L.nq, n), generator=g) * 2 - 1) * s
        if qvel_noise is None:
            qvel_noise = (torch.rand((L.nv, n), generator=g) * 2 - 1) * s
        ci = torch.as_tensor(clip_idx, dtype=torch.int32).to(dev).contiguous()
        sf = torch.as_tensor(start_frame, dtype=torch.int32).to(dev).contiguous()
        qn = torch.as_tensor(qpos_noise, dtype=torch.float32).to(dev).contiguous()
        vn = torch.as_tensor(qvel_noise, dtype=torch.float32).to(dev).contiguous()
        if ci.shape != (n,) or sf.shape != (n,) or qn.shape != (L.nq, n) or vn.shape != (L.nv, n):
            raise ValueError("reset inputs have the wrong shape")
        with torch.cuda.device(dev):
            _hip.check(self._L.tmjx_reset(self._handle, _ptr(self.state_buf), _ptr(self.istate_buf), _ptr(ci), _ptr(sf), _ptr(qn),
                                          _ptr(vn), _ptr(self.obs_buf), _ptr(self.workspace), n, self._stream()), "tmjx_reset")
        self.reward_buf.zero_(); self.done_buf.zero_(); self.trunc_buf.zero_(); self.metrics_buf.zero_()
        self._keep = (ci, sf, qn, vn)  # keep inputs alive until the stream has consumed them
        return self._state()

    def step(self, state: State | None, action: torch.Tensor) -> State:
        """step(state, action) (reference: task/single_clip_tracking.py:207). `action` is [n, nu] or [nu][n]."""
        n, L = self.num_envs, self.layout
        if action.shape == (n, L.nu) and not (n == L.nu and action.stride(0) == 1):
            a = action.t().contiguous()
        elif action.shape == (L.nu, n):
            a = action.contiguous()
        else:
            raise ValueError(f"action must be [{n},{L.nu}] or [{L.nu},{n}]")
        a = a.to(device=self.device, dtype=torch.float32)
        if self._physics_events is not None:
            if getattr(self, "_action_repeat", 1) != 1:
                raise RuntimeError("the K2-bracketing measurement mode steps the physics once: action_repeat must be 1")
            # measurement mode (bench.py): the same kernels as tmjx_step, issued as K2 then K3 so that HIP events on the
            # launch stream bracket the physics kernel alone
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.device(self.device):
                e0.record(torch.cuda.current_stream(self.device))
                _hip.check(self._L.tmjx_physics_step(self._handle, _ptr(self.state_buf), _ptr(a), _ptr(self.workspace), n, self._stream()), "tmjx_physics_step")
                e1.record(torch.cuda.current_stream(self.device))
                _hip.check(self._L.tmjx_reward_obs(self._handle, _ptr(self.state_buf), _ptr(self.istate_buf), _ptr(a), _ptr(self.obs_buf),
                                                   _ptr(self.reward_buf), _ptr(self.done_buf), _ptr(self.trunc_buf), _ptr(self.metrics_buf),
                                                   _ptr(self.workspace), n, self._stream()), "tmjx_reward_obs")
            self._physics_events.append((e0, e1))
            self._keep_a = a
            return self._state()
        with torch.cuda.device(self.device):
            _hip.check(self._L.tmjx_step(self._handle, _ptr(self.state_buf), _ptr(self.istate_buf), _ptr(a), _ptr(self.obs_buf),
                                         _ptr(self.reward_buf), _ptr(self.done_buf), _ptr(self.trunc_buf), _ptr(self.metrics_buf),
                                         _ptr(self.workspace), n, self._stream()), "tmjx_step")
        self._keep_a = a
        return self._state()

    # ---- K2 / K3 alone (tests, profiling)
    def physics(self, action_rows: torch.Tensor | None, n_substeps: int) -> None:
        with torch.cuda.device(self.device):
            _hip.check(self._L.tmjx_physics(self._handle, _ptr(self.state_buf), _ptr(action_rows), int(n_substeps),
                                            _ptr(self.workspace), self.num_envs, self._stream()), "tmjx_physics")

    def forward(self) -> None:
        with torch.cuda.device(self.device):
            _hip.check(self._L.tmjx_forward(self._handle, _ptr(self.state_buf), _ptr(self.workspace), self.num_envs, self._stream()),
                       "tmjx_forward")

    def reward_obs(self, action_rows: torch.Tensor) -> State:
        with torch.cuda.device(self.device):
            _hip.check(self._L.tmjx_reward_obs(self._handle, _ptr(self.state_buf), _ptr(self.istate_buf), _ptr(action_rows),
                                               _ptr(self.obs_buf), _ptr(self.reward_buf), _ptr(self.done_buf), _ptr(self.trunc_buf),
                                               _ptr(self.metrics_buf), _ptr(self.workspace), self.num_envs, self._stream()), "tmjx_reward_obs")
        return self._state()

    def rows(self, name: str) -> torch.Tensor:
        """Named per-env array ([count][n] view) of the state or workspace buffer (debug/tests)."""
        r0, cnt = C.c_int32(), C.c_int32()
        k = self._L.tmjx_debug_rows(self._handle, name.encode(), C.byref(r0), C.byref(cnt))
        if k < 0:
            raise KeyError(name)
        buf = self.state_buf if k == 1 else self.workspace
        return buf[r0.value:r0.value + cnt.value]

    # ---- reference helpers used by callers
    def _get_cur_frame(self) -> torch.Tensor:
        L = self.layout
        t = self.state_buf[L.time] * float(self._mocap_hz)
        return torch.floor(t + self.istate_buf[L.i_start_frame].float()).to(torch.int32)

    def configure_wrappers(self, episode_length: int, auto_reset: bool, action_repeat: int = 1) -> None:
        """Switch the Episode / AutoReset wrapper semantics of the handle (wrappers.wrap): two constants of the device model change,
        the clip table stays resident (tmjx_set_wrappers); `action_repeat` is brax EpisodeWrapper's (tmjx_set_action_repeat)."""
        if int(action_repeat) < 1:
            raise ValueError("action_repeat must be >= 1")
        self._episode_length, self._auto_reset, self._action_repeat = int(episode_length), bool(auto_reset), int(action_repeat)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)        # no launch of this handle may be in flight while its constants change
            _hip.check(self._L.tmjx_set_wrappers(self._handle, self._episode_length, int(self._auto_reset)), "tmjx_set_wrappers")
            _hip.check(self._L.tmjx_set_action_repeat(self._handle, self._action_repeat), "tmjx_set_action_repeat")
        self._blob = build_blob(self.walker, n_frames=self._n_frames, mocap_hz=self._mocap_hz, clip_length=self._clip_length,
                                traj_length=self._ref_len, window=int(self._reward_config.var_window_size),
                                episode_length=self._episode_length, reward_f=self._reward_config.vector(),
                                auto_reset=self._auto_reset, **self._opts)      # what a handle with these semantics is created from (tests: the oracle)
