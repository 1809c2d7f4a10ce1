"""Reference-clip loading — mirror of track_mjx/io/load.py:61-278 on the numpy HDF5 reader (h5lite.py; h5py is not in the image).

Same function names, arguments, return layout (leaves shaped (clips, frames, dims...)) and error behaviour:
`load_data` tries the stac-mjx layout first (qpos / qvel / xpos / xquat with all clips' frames back to back + a `config`
YAML string holding stac.n_frames_per_clip) and falls back to the ReferenceClip layout (group `all_clips` with the eight
leaves) on KeyError, exactly as load.py:61-76 does.  Arrays are float32 numpy (the reference converts to jax float32 arrays)."""
from __future__ import annotations

import re
from typing import Tuple

import numpy as np
import yaml

from .. import h5lite
from ..clips import ReferenceClip

_LEAVES = ("angular_velocity", "body_positions", "body_quaternions", "joints", "joints_velocity", "position", "quaternion", "velocity")


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _config_of(data) -> dict:
    raw = data["config"][()]
    if isinstance(raw, np.ndarray):
        raw = raw.item()
    return yaml.safe_load(raw.decode("utf-8") if isinstance(raw, (bytes, np.bytes_)) else str(raw))


def load_data(data_path: str) -> ReferenceClip:
    """load.py:61-76."""
    try:
        return make_multiclip_data(data_path)
    except KeyError:
        return load_reference_clip_data(data_path)


def make_singleclip_data(traj_data_path):
    """load.py:79-103: one clip, leaves shaped (frames, dims...); returned as a 1-tuple like the reference does."""
    with h5lite.File(traj_data_path, "r") as data:
        qpos, qvel, xpos, xquat = (_f32(data[k][()]) for k in ("qpos", "qvel", "xpos", "xquat"))
    return (ReferenceClip(position=qpos[:, :3], quaternion=qpos[:, 3:7], joints=qpos[:, 7:], body_positions=xpos, velocity=qvel[:, :3],
                          angular_velocity=qvel[:, 3:6], joints_velocity=qvel[:, 6:], body_quaternions=xquat),)


def make_multiclip_data(traj_data_path, n_frames_per_clip: int | None = None) -> ReferenceClip:
    """load.py:105-137: (clips * frames, dims...) -> (clips, frames, dims...)."""
    def reshape_frames(arr, clip_len):
        a = arr[()]
        return _f32(a.reshape(a.shape[0] // clip_len, clip_len, *a.shape[1:]))

    with h5lite.File(traj_data_path, "r") as data:
        if n_frames_per_clip is None:
            n_frames_per_clip = _config_of(data)["stac"]["n_frames_per_clip"]
        q, x, v, xq = (reshape_frames(data[k], n_frames_per_clip) for k in ("qpos", "xpos", "qvel", "xquat"))
    return ReferenceClip(position=q[:, :, :3], quaternion=q[:, :, 3:7], joints=q[:, :, 7:], body_positions=x, velocity=v[:, :, :3],
                         angular_velocity=v[:, :, 3:6], joints_velocity=v[:, :, 6:], body_quaternions=xq)


def load_reference_clip_data(filepath: str, group_name: str = "all_clips") -> ReferenceClip:
    """load.py:140-183."""
    try:
        with h5lite.File(filepath, "r") as f:
            if group_name not in f:
                raise KeyError(f"Group '{group_name}' not found in the HDF5 file.")
            group = f[group_name]
            out = {}
            for key in _LEAVES:
                if key not in group:
                    raise KeyError(f"Dataset '{key}' not found in group '{group_name}'.")
                out[key] = _f32(group[key][()])
            return ReferenceClip(**out)
    except FileNotFoundError:
        raise FileNotFoundError(f"File not found: {filepath}")
    except h5lite.H5Error as e:
        raise OSError(f"Error reading HDF5 file: {filepath} - {e}")


def select_clips(clips: ReferenceClip, indices) -> ReferenceClip:
    """load.py:258-278."""
    indices = np.array(indices)
    return ReferenceClip(**{k: getattr(clips, k)[indices] for k in _LEAVES}, original_clip_idx=indices[:, np.newaxis])


def generate_train_test_split(data: ReferenceClip, test_ratio: float = 0.1) -> Tuple[ReferenceClip, ReferenceClip]:
    """load.py:186-211 (numpy's global RandomState, as the reference: seed it with np.random.seed for a reproducible split)."""
    num_clips = data.position.shape[0]
    indices = np.arange(num_clips)
    test_idx = np.random.choice(indices, size=int(num_clips * test_ratio), replace=False)
    train_idx = indices[~np.isin(indices, test_idx)]
    train_idx.sort()
    test_idx.sort()
    return select_clips(data, train_idx), select_clips(data, test_idx)


def load_clips_metadata(traj_data_path: str) -> list:
    """load.py:214-240: (behaviour name, number) per clip from config.model.snips_order."""
    with h5lite.File(traj_data_path, "r") as data:
        config = _config_of(data)
    pattern = re.compile(r"/([^/]+)_([0-9]+)\.p$")
    out = []
    for path in config["model"]["snips_order"]:
        m = pattern.search(path)
        if m:
            out.append((m.group(1), int(m.group(2))))
    return out


def sub_sample_training_set(train_idx: np.ndarray, train_ratio: float = 0.1):
    """load.py:243-255."""
    sampled = np.random.choice(train_idx, size=int(len(train_idx) * train_ratio), replace=False)
    sampled.sort()
    return sampled
