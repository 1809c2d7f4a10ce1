"""Reference clips: the `ReferenceClip` table layout and the synthetic clip generator.

Layout mirrors the reference pytree (track_mjx/io/load.py:16-38, leaves shaped
(clips, frames, dims...), load.py:105-137).  No data file ships with the reference and h5py is
absent, so benchmark/test inputs are synthetic: sinusoidal joints inside the
model's joint ranges, slow forward root translation with a small yaw, body positions from this
module's own batched forward kinematics stored with the reference's `xpos[1:]` row alignment
(67 rows: row i = body i+1), velocities by finite differences at mocap_hz.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

FIELDS = ("position", "quaternion", "joints", "body_positions", "velocity", "angular_velocity",
          "joints_velocity", "body_quaternions")


@dataclass
class ReferenceClip:
    position: np.ndarray          # (C, F, 3)
    quaternion: np.ndarray        # (C, F, 4)
    joints: np.ndarray            # (C, F, nq-7)
    body_positions: np.ndarray    # (C, F, nbody-1, 3)
    velocity: np.ndarray          # (C, F, 3)
    angular_velocity: np.ndarray  # (C, F, 3)
    joints_velocity: np.ndarray   # (C, F, nv-6)
    body_quaternions: np.ndarray  # (C, F, nbody-1, 4)
    original_clip_idx: np.ndarray | None = None

    def as_dict(self):
        return {k: getattr(self, k) for k in FIELDS}


def _qmul(a, b):
    w = a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3]
    x = a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2]
    y = a[..., 0] * b[..., 2] - a[..., 1] * b[..., 3] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1]
    z = a[..., 0] * b[..., 3] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1] + a[..., 3] * b[..., 0]
    return np.stack([w, x, y, z], axis=-1)


def _qrot(q, v):
    s, u = q[..., :1], q[..., 1:]
    return 2 * np.sum(u * v, -1, keepdims=True) * u + (s * s - np.sum(u * u, -1, keepdims=True)) * v + 2 * s * np.cross(u, v)


def batched_fk(model: dict, qpos: np.ndarray):
    """qpos (N, nq) float64 -> xpos (N, nbody, 3), xquat (N, nbody, 4). Free joint + hinges only."""
    nbody, njnt = int(model["dims"][0]), int(model["dims"][1])
    N = qpos.shape[0]
    body_pos = model["body_pos"].reshape(nbody, 3)
    body_quat = model["body_quat"].reshape(nbody, 4)
    jnt_pos = model["jnt_pos"].reshape(njnt, 3)
    jnt_axis = model["jnt_axis"].reshape(njnt, 3)
    xpos = np.zeros((N, nbody, 3))
    xquat = np.zeros((N, nbody, 4))
    xquat[:, 0, 0] = 1.0
    for b in range(1, nbody):
        p = int(model["body_parentid"][b])
        pos = xpos[:, p] + _qrot(xquat[:, p], np.broadcast_to(body_pos[b], (N, 3)))
        quat = _qmul(xquat[:, p], np.broadcast_to(body_quat[b], (N, 4)))
        for j in range(int(model["body_jntadr"][b]), int(model["body_jntadr"][b]) + int(model["body_jntnum"][b])):
            a = int(model["jnt_qposadr"][j])
            if int(model["jnt_type"][j]) == 0:
                pos = qpos[:, a:a + 3]
                quat = qpos[:, a + 3:a + 7] / np.linalg.norm(qpos[:, a + 3:a + 7], axis=-1, keepdims=True)
            else:
                anchor = _qrot(quat, np.broadcast_to(jnt_pos[j], (N, 3))) + pos
                ang = (qpos[:, a] - model["qpos0"][a]) * 0.5
                qloc = np.concatenate([np.cos(ang)[:, None], np.sin(ang)[:, None] * jnt_axis[j][None, :]], axis=-1)
                quat = _qmul(quat, qloc)
                pos = anchor - _qrot(quat, np.broadcast_to(jnt_pos[j], (N, 3)))
        xpos[:, b], xquat[:, b] = pos, quat
    return xpos, xquat


def make_synthetic_clips(model: dict, n_clips: int, n_frames: int = 250, seed: int = 0, mocap_hz: int = 50) -> ReferenceClip:
    rng = np.random.default_rng(seed)
    nbody, njnt, nq, nv = (int(model["dims"][i]) for i in range(4))
    rngs = model["jnt_range"].reshape(njnt, 2)[1:]
    mid, half = 0.5 * (rngs[:, 0] + rngs[:, 1]), 0.5 * (rngs[:, 1] - rngs[:, 0])
    f = np.arange(n_frames)
    k = rng.integers(1, 4, size=(n_clips, nq - 7))
    phi = rng.uniform(0, 2 * np.pi, size=(n_clips, nq - 7))
    joints = mid + 0.25 * half * np.sin(2 * np.pi * f[None, :, None] / n_frames * k[:, None, :] + phi[:, None, :])
    yaw = 0.2 * np.sin(2 * np.pi * f / n_frames)
    quat = np.zeros((n_clips, n_frames, 4))
    quat[..., 0], quat[..., 3] = np.cos(yaw / 2), np.sin(yaw / 2)
    pos = np.zeros((n_clips, n_frames, 3))
    pos[..., 0] = 0.1 * f / n_frames
    qpos = np.concatenate([pos, quat, joints], axis=-1).reshape(-1, nq)
    xpos, xquat = batched_fk(model, qpos)
    # standing height: lowest paw-capsule end of the whole clip rests 1 mm above the floor plane
    g2b = model["con_body2"]
    gpos = model["con_g2_pos"].reshape(-1, 3)
    gsize = model["con_g2_size"].reshape(-1, 3)
    plane_z = float(model["con_g1_pos"].reshape(-1, 3)[0, 2])
    low = np.full((n_clips, n_frames), np.inf)
    for c in range(len(g2b)):
        b = int(g2b[c])
        p = xpos[:, b] + _qrot(xquat[:, b], np.broadcast_to(gpos[c], (qpos.shape[0], 3)))
        low = np.minimum(low, (p[:, 2] - gsize[c, :2].max() - gsize[c, 0]).reshape(n_clips, n_frames))
    z0 = plane_z + 1e-3 - low.min(axis=1)
    pos[..., 2] = z0[:, None]
    xpos = xpos.reshape(n_clips, n_frames, nbody, 3)
    xpos[..., 2] += z0[:, None, None]
    xquat = xquat.reshape(n_clips, n_frames, nbody, 4)

    def fd(x):
        v = np.zeros_like(x)
        v[:, :-1] = (x[:, 1:] - x[:, :-1]) * mocap_hz
        v[:, -1] = v[:, -2]
        return v

    angvel = np.zeros((n_clips, n_frames, 3))
    angvel[..., 2] = fd(yaw[None, :].repeat(n_clips, 0))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return ReferenceClip(
        position=f32(pos), quaternion=f32(quat), joints=f32(joints), body_positions=f32(xpos[:, :, 1:]),
        velocity=f32(fd(pos)), angular_velocity=f32(angvel), joints_velocity=f32(fd(joints)),
        body_quaternions=f32(xquat[:, :, 1:]), original_clip_idx=np.arange(n_clips, dtype=np.int32))
