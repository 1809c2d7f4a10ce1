"""Rodent walker: host-side mirror of the reference walker interface.

Reference: track_mjx/environment/walker/rodent.py:16-114 (constructor arguments, name -> id
tables) and walker/base.py:70-88 (index properties).  The MuJoCo compile step is replaced by
the pre-compiled model blob `assets/rodent_model.tmjx` (tools/compile_model.py), which already
contains the torque-actuator rewrite and the 0.9 rescale the reference config asks for.
"""
from __future__ import annotations

from collections import OrderedDict
from pathlib import Path
from typing import Sequence

import numpy as np

from . import blob as _blob

_ASSETS = Path(__file__).parent / "assets"


class Rodent:
    def __init__(self, joint_names: Sequence[str], body_names: Sequence[str], end_eff_names: Sequence[str],
                 *, torque_actuators: bool = False, rescale_factor: float = 0.9):
        if not torque_actuators or abs(rescale_factor - 0.9) > 1e-12:
            raise NotImplementedError(
                "the shipped model blob is compiled for torque_actuators=True, rescale_factor=0.9 "
                "(rodent-full-clips.yaml:116-117); recompile with tools/compile_model.py for other values")
        self._torso_name = "torso"
        self._joint_names = list(joint_names)
        self._body_names = list(body_names)
        self._end_eff_names = list(end_eff_names)
        self.model = _blob.load(_ASSETS / "rodent_model.tmjx")
        self.names = {"body": {}, "joint": {}, "actuator": {}}
        with open(_ASSETS / "rodent_model.names.txt") as f:
            for line in f:
                kind, idx, name = line.split()
                self.names[kind][name] = int(idx)
        dims = self.model["dims"]
        self.nbody, self.njnt, self.nq, self.nv, self.nu, self.ncon = (int(x) for x in dims)
        self._initialize_indices()

    def _initialize_indices(self) -> None:
        self._joint_idxs = np.array([self.names["joint"][j] for j in self._joint_names], dtype=np.int32)
        self._body_idxs = np.array([self.names["body"][b] for b in self._body_names], dtype=np.int32)
        self._endeff_idxs = np.array([self.names["body"][e] for e in self._end_eff_names], dtype=np.int32)
        self._torso_idx = int(self.names["body"][self._torso_name])

    joint_idxs = property(lambda self: self._joint_idxs)
    body_idxs = property(lambda self: self._body_idxs)
    endeff_idxs = property(lambda self: self._endeff_idxs)
    torso_idx = property(lambda self: self._torso_idx)


def build_blob(walker: Rodent, *, n_frames: int, iterations: int, ls_iterations: int, timestep: float,
               mocap_hz: int, clip_length: int, traj_length: int, window: int, episode_length: int,
               reward_f: np.ndarray, tolerance: float = 1e-8, ls_tolerance: float = 0.01,
               impratio: float = 1.0, auto_reset: bool = True) -> bytes:
    """Model constants + env/task configuration -> the blob `tmjx_model_create` consumes."""
    e = OrderedDict(walker.model)
    e["opt_f"] = np.array([timestep, tolerance, ls_tolerance, impratio], dtype=np.float64)
    e["opt_i"] = np.array([iterations, ls_iterations, n_frames], dtype=np.int32)
    e["env_i"] = np.array([mocap_hz, clip_length, traj_length, window, walker.torso_idx, episode_length, int(auto_reset)], dtype=np.int32)
    e["joint_idxs"] = walker.joint_idxs
    e["body_idxs"] = walker.body_idxs
    e["endeff_idxs"] = walker.endeff_idxs
    e["reward_f"] = np.asarray(reward_f, dtype=np.float64)
    return _blob.pack(e)
