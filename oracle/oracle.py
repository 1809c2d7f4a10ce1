"""ctypes binding of the CPU oracle (test infrastructure only — never imported by the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_DP = C.POINTER(C.c_double)
_FP = C.POINTER(C.c_float)


def build() -> None:
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)


def _dp(a):
    return a.ctypes.data_as(_DP)


class Oracle:
    """One oracle model (+ clips) and helpers to run single envs."""

    def __init__(self, blob: bytes, precision: str = "f32"):
        so = _HERE / f"liboracle_{precision}.so"
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        self.L = L
        L.oracle_model_create.restype = C.c_void_p
        L.oracle_model_create.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_model_destroy.argtypes = [C.c_void_p]
        L.oracle_sizeof_data.restype = C.c_size_t
        L.oracle_sizeof_env.restype = C.c_size_t
        L.oracle_set_clips.argtypes = [C.c_void_p, _FP, _FP, _FP, _FP, _FP, C.c_int, C.c_int]
        L.oracle_data_init.argtypes = [C.c_void_p, C.c_void_p, _DP, _DP]
        L.oracle_forward.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_step.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_ctrl.argtypes = [C.c_void_p, C.c_void_p, _DP]
        L.oracle_data_get.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _DP, C.c_int]
        L.oracle_data_set.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _DP, C.c_int]
        L.oracle_env_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _DP, _DP]
        L.oracle_env_step.argtypes = [C.c_void_p, C.c_void_p, _DP]
        L.oracle_env_step_ex.argtypes = [C.c_void_p, C.c_void_p, _DP, C.c_int]
        L.oracle_set_action_repeat.argtypes = [C.c_void_p, C.c_int]
        L.oracle_env_set.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _DP, C.c_int]
        L.oracle_env_get.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _DP, C.c_int]
        L.oracle_obs_size.argtypes = [C.c_void_p]
        L.oracle_env_step_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, _DP, C.c_int]
        L.oracle_gae.argtypes = [_DP, _DP, _DP, _DP, _DP, C.c_double, C.c_double, _DP, _DP, C.c_int, C.c_int]
        self._blob = blob
        self.m = L.oracle_model_create(blob, len(blob))
        if not self.m:
            raise RuntimeError("oracle_model_create failed")
        self.sizeof_data = L.oracle_sizeof_data()
        self.sizeof_env = L.oracle_sizeof_env()
        self._keep = []

    def __del__(self):
        try:
            self.L.oracle_model_destroy(self.m)
        except Exception:
            pass

    # ---- clips
    def set_clips(self, clips: dict) -> None:
        arrs = [np.ascontiguousarray(clips[k], dtype=np.float32)
                for k in ("position", "quaternion", "joints", "body_positions", "angular_velocity")]
        n_clips, n_frames = arrs[0].shape[:2]
        self.L.oracle_set_clips(self.m, *[a.ctypes.data_as(_FP) for a in arrs], n_clips, n_frames)

    # ---- raw physics data
    def new_data(self, qpos, qvel):
        buf = C.create_string_buffer(self.sizeof_data)
        q = np.ascontiguousarray(qpos, dtype=np.float64)
        v = np.ascontiguousarray(qvel, dtype=np.float64)
        self.L.oracle_data_init(self.m, buf, _dp(q), _dp(v))
        return buf

    def forward(self, d):
        self.L.oracle_forward(self.m, d)

    def step(self, d, ctrl=None):
        if ctrl is not None:
            c = np.ascontiguousarray(ctrl, dtype=np.float64)
            self.L.oracle_set_ctrl(self.m, d, _dp(c))
        self.L.oracle_step(self.m, d)

    def get(self, d, name, cap=80 * 270):
        out = np.zeros(cap, dtype=np.float64)
        n = self.L.oracle_data_get(self.m, d, name.encode(), _dp(out), cap)
        if n < 0:
            raise KeyError(name)
        return out[:n].copy()

    def set(self, d, name, val):
        v = np.ascontiguousarray(val, dtype=np.float64).ravel()
        if self.L.oracle_data_set(self.m, d, name.encode(), _dp(v), v.size) < 0:
            raise KeyError(name)

    # ---- envs
    def new_envs(self, n):
        return C.create_string_buffer(self.sizeof_env * n)

    def env_ptr(self, envs, i):
        return C.c_void_p(C.addressof(envs) + i * self.sizeof_env)

    def env_reset(self, envs, i, clip_idx, start_frame, qpos_noise, qvel_noise):
        qn = np.ascontiguousarray(qpos_noise, dtype=np.float64)
        vn = np.ascontiguousarray(qvel_noise, dtype=np.float64)
        self.L.oracle_env_reset(self.m, self.env_ptr(envs, i), int(clip_idx), int(start_frame), _dp(qn), _dp(vn))

    def env_step(self, envs, i, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        self.L.oracle_env_step(self.m, self.env_ptr(envs, i), _dp(a))

    def set_action_repeat(self, action_repeat):
        """brax EpisodeWrapper's repeat count for env_step (wrappers.py:43)."""
        self.L.oracle_set_action_repeat(self.m, int(action_repeat))

    def env_post(self, envs, i, action):
        """K3 alone: everything in env.step except the physics substeps."""
        a = np.ascontiguousarray(action, dtype=np.float64)
        self.L.oracle_env_step_ex(self.m, self.env_ptr(envs, i), _dp(a), 0)

    def env_set(self, envs, i, name, val):
        v = np.ascontiguousarray(val, dtype=np.float64).ravel()
        if self.L.oracle_env_set(self.m, self.env_ptr(envs, i), name.encode(), _dp(v), v.size) < 0:
            raise KeyError(name)

    def env_step_batch(self, envs, n, actions, nthreads=0):
        a = np.ascontiguousarray(actions, dtype=np.float64)
        self.L.oracle_env_step_batch(self.m, envs, n, _dp(a), nthreads)

    def env_get(self, envs, i, name, cap=80 * 270):
        out = np.zeros(cap, dtype=np.float64)
        n = self.L.oracle_env_get(self.m, self.env_ptr(envs, i), name.encode(), _dp(out), cap)
        if n < 0:
            raise KeyError(name)
        return out[:n].copy()

    def obs_size(self):
        return self.L.oracle_obs_size(self.m)

    def gae(self, truncation, termination, rewards, values, bootstrap, lambda_, discount):
        T, B = rewards.shape
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (truncation, termination, rewards, values, bootstrap)]
        vs = np.zeros((T, B))
        adv = np.zeros((T, B))
        self.L.oracle_gae(*[_dp(a) for a in arrs], lambda_, discount, _dp(vs), _dp(adv), T, B)
        return vs, adv
