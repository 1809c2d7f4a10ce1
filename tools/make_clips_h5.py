#!/usr/bin/env python3
"""Export synthetic reference clips as an HDF5 file in the stac-mjx layout the reference loads (track_mjx/io/load.py:105-137):
qpos / qvel / xpos / xquat with the frames of all clips back to back + a `config` YAML string (stac.n_frames_per_clip,
model.snips_order).  Written with track_mjx_amd.h5lite.write_file (no h5py needed); readable by h5py / libhdf5.

    python tools/make_clips_h5.py --clips 64 --out /tmp/clips.h5
    python -m track_mjx_amd.train data_path=/tmp/clips.h5 train_setup.train_config.num_envs=4096
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from track_mjx_amd import clips as _clips, config as _config, h5lite  # noqa: E402
from track_mjx_amd.walker import Rodent  # noqa: E402


def export(clip: _clips.ReferenceClip, path) -> None:
    C, F = clip.position.shape[:2]
    qpos = np.concatenate([clip.position, clip.quaternion, clip.joints], axis=-1).reshape(C * F, -1)
    qvel = np.concatenate([clip.velocity, clip.angular_velocity, clip.joints_velocity], axis=-1).reshape(C * F, -1)
    cfg = "stac:\n  n_frames_per_clip: %d\nmodel:\n  snips_order:\n%s" % (F, "".join(f"  - /synthetic/Synth_{i}.p\n" for i in range(C)))
    h5lite.write_file(path, {"qpos": qpos.astype(np.float32), "qvel": qvel.astype(np.float32),
                             "xpos": clip.body_positions.reshape(C * F, -1, 3).astype(np.float32),
                             "xquat": clip.body_quaternions.reshape(C * F, -1, 4).astype(np.float32), "config": cfg.encode("utf-8")})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    cfg = _config.default_config()
    export(_clips.make_synthetic_clips(Rodent(**cfg["walker_config"]).model, a.clips, n_frames=cfg["reference_config"]["clip_length"], seed=a.seed), a.out)
    print("wrote", a.out)
