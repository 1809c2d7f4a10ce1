"""ctypes wrapper of the TEST-ONLY host emulation of the HIP kernel bodies (tests/hostemu/hostemu.cpp)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_FP, _IP = C.POINTER(C.c_float), C.POINTER(C.c_int)


def build():
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", str(_HERE / "libhostemu.so"),
                    str(_HERE / "hostemu.cpp")], check=True, capture_output=True)


def _f(a):
    return a.ctypes.data_as(_FP)


def _i(a):
    return a.ctypes.data_as(_IP)


class Emu:
    def __init__(self, blob: bytes, n_env: int):
        src = [_HERE / "hostemu.cpp", *sorted((_HERE.parents[1] / "track_mjx_amd" / "csrc").glob("*.h"))]
        so = _HERE / "libhostemu.so"
        if not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in src):
            build()
        self.L = L = C.CDLL(str(so))
        L.emu_model_create.restype = C.c_void_p
        L.emu_model_create.argtypes = [C.c_char_p, C.c_size_t]
        L.emu_last_error.restype = C.c_char_p
        self.m = C.c_void_p(L.emu_model_create(blob, len(blob)))
        if not self.m:
            raise RuntimeError(L.emu_last_error().decode())
        lay = (C.c_int * 5)()
        L.emu_layout(self.m, lay)
        self.s_rows, self.i_rows, self.w_rows, self.obs_size, self.nphys = list(lay)
        self.n = n_env
        self.st = np.zeros((self.s_rows, n_env), np.float32)
        self.ist = np.zeros((self.i_rows, n_env), np.int32)
        self.ws = np.zeros((self.w_rows, n_env), np.float32)
        self.obs = np.zeros((self.obs_size, n_env), np.float32)
        self.reward = np.zeros(n_env, np.float32)
        self.done = np.zeros(n_env, np.float32)
        self.trunc = np.zeros(n_env, np.float32)
        self.metrics = np.zeros((20, n_env), np.float32)

    def rows(self, name):
        r0, cnt = C.c_int(), C.c_int()
        k = self.L.emu_rows(self.m, name.encode(), C.byref(r0), C.byref(cnt))
        if k < 0:
            raise KeyError(name)
        buf = self.st if k == 1 else self.ws
        return buf[r0.value:r0.value + cnt.value]

    def set_clips(self, clips: dict):
        arrs = [np.ascontiguousarray(clips[k], dtype=np.float32)
                for k in ("position", "quaternion", "joints", "body_positions", "angular_velocity")]
        nc, nf = arrs[0].shape[:2]
        self.L.emu_clips(self.m, *[_f(a) for a in arrs], nc, nf)

    def reset(self, clip_idx, start_frame, qn, vn):
        c = np.ascontiguousarray(clip_idx, np.int32)
        s = np.ascontiguousarray(start_frame, np.int32)
        qn = np.ascontiguousarray(qn, np.float32)
        vn = np.ascontiguousarray(vn, np.float32)
        self.L.emu_reset(self.m, _f(self.st), _i(self.ist), _i(c), _i(s), _f(qn), _f(vn), _f(self.obs), _f(self.ws), self.n)

    def step(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.emu_step(self.m, _f(self.st), _i(self.ist), _f(a), _f(self.obs), _f(self.reward), _f(self.done),
                        _f(self.trunc), _f(self.metrics), _f(self.ws), self.n)

    def step_repeat(self, action, action_repeat):
        """step with brax EpisodeWrapper's action_repeat (tmjx_set_action_repeat)."""
        a = np.ascontiguousarray(action, np.float32)
        self.L.emu_step_repeat(self.m, _f(self.st), _i(self.ist), _f(a), _f(self.obs), _f(self.reward), _f(self.done),
                               _f(self.trunc), _f(self.metrics), _f(self.ws), self.n, int(action_repeat))

    def physics(self, action, nsub, do_euler=True):
        a = None if action is None else np.ascontiguousarray(action, np.float32)
        self.L.emu_physics(self.m, _f(self.st), _f(a) if a is not None else None, nsub, int(do_euler), _f(self.ws), self.n)

    def physics_wave(self, action, nsub, do_euler=True, dump=True):
        """The wave-per-env kernel body (64 emulated lanes, LDS image per env)."""
        a = None if action is None else np.ascontiguousarray(action, np.float32)
        self.L.emu_physics_wave(self.m, _f(self.st), _f(a) if a is not None else None, nsub, int(do_euler),
                                _f(self.ws) if dump else None, self.n)

    def lds_bytes(self):
        return 4 * self.L.emu_lds_floats(self.m)

    def post(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.emu_post(self.m, _f(self.st), _i(self.ist), _f(a), _f(self.obs), _f(self.reward), _f(self.done),
                        _f(self.trunc), _f(self.metrics), self.n)
