#!/usr/bin/env python3
"""Writes the HDF5 fixtures of tests/test_h5lite.py with the REAL h5py (libhdf5) — run with an interpreter that has h5py
(this image: /opt/conda/bin/python3.9, h5py 3.3.0 / libhdf5 1.10.6; the product and the tests never import h5py):

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py

Files (a few KB each) + h5_expected.npz holding the arrays that were written:
  stac_small.h5        stac-mjx layout the reference reads (track_mjx/io/load.py:105-137): qpos / qvel / xpos / xquat with frames
                       of all clips back to back, `config` = YAML string scalar (variable-length UTF-8, as h5py stores a str);
                       one dataset per storage kind: contiguous f32, contiguous f64, chunked + gzip + shuffle, chunked raw
  refclip_small.h5     ReferenceClip layout (load.py:140-183): group `all_clips` with the eight leaves; fixed-length config string
  stac_latest.h5       the stac layout written with libver="latest" (superblock v3, version-2 object headers, link messages)
"""
from pathlib import Path

import h5py
import numpy as np

HERE = Path(__file__).resolve().parent
rng = np.random.default_rng(123)
n_clips, clip_len, nq, nv, nb = 3, 10, 74, 73, 67
F = n_clips * clip_len
qpos = rng.normal(size=(F, nq)).astype(np.float32)
qvel = rng.normal(size=(F, nv)).astype(np.float64)
xpos = rng.normal(size=(F, nb, 3)).astype(np.float32)
xquat = rng.normal(size=(F, nb, 4)).astype(np.float32)
config = ("stac:\n  n_frames_per_clip: 10\nmodel:\n  snips_order:\n  - /data/snips/Walk_12.p\n  - /data/snips/Rear_3.p\n"
          "  - /data/snips/FaceGroom_7.p\n# é non-ascii survives\n")


def write_stac(path, **kw):
    with h5py.File(path, "w", **kw) as f:
        f.create_dataset("qpos", data=qpos)
        f.create_dataset("qvel", data=qvel)
        f.create_dataset("xpos", data=xpos, chunks=(8, nb, 3), compression="gzip", compression_opts=4, shuffle=True)
        f.create_dataset("xquat", data=xquat, chunks=(7, 16, 4))
        f.create_dataset("config", data=config)
        f.create_dataset("kp_names", data=np.array([b"Snout", b"EarL", b"EarR"]))      # fixed-length byte strings
        f.create_dataset("offsets", data=np.arange(12, dtype=np.int64).reshape(3, 4))
        f.attrs["note"] = "fixture"


write_stac(HERE / "stac_small.h5")
write_stac(HERE / "stac_latest.h5", libver="latest")

leaves = {
    "position": qpos[:, :3].reshape(n_clips, clip_len, 3), "quaternion": qpos[:, 3:7].reshape(n_clips, clip_len, 4),
    "joints": qpos[:, 7:].reshape(n_clips, clip_len, nq - 7), "body_positions": xpos.reshape(n_clips, clip_len, nb, 3),
    "velocity": qvel[:, :3].reshape(n_clips, clip_len, 3).astype(np.float32),
    "angular_velocity": qvel[:, 3:6].reshape(n_clips, clip_len, 3).astype(np.float32),
    "joints_velocity": qvel[:, 6:].reshape(n_clips, clip_len, nv - 6).astype(np.float32),
    "body_quaternions": xquat.reshape(n_clips, clip_len, nb, 4),
}
with h5py.File(HERE / "refclip_small.h5", "w") as f:
    g = f.create_group("all_clips")
    for i, (k, v) in enumerate(leaves.items()):
        if i % 2:
            g.create_dataset(k, data=v, compression="gzip")
        else:
            g.create_dataset(k, data=v)
    f.create_dataset("config", data=np.bytes_(config.encode("utf-8")))
    sub = f.create_group("meta/inner")
    sub.create_dataset("scalar", data=np.float64(2.5))
    many = f.create_group("many")          # > 256 links: a two-level group B-tree with many symbol-table nodes
    for i in range(270):
        many.create_dataset(f"d{i:03d}", data=np.int16(i))

np.savez_compressed(HERE / "h5_expected.npz", qpos=qpos, qvel=qvel, xpos=xpos, xquat=xquat,
                    config=np.frombuffer(config.encode("utf-8"), dtype=np.uint8), **{"leaf_" + k: v for k, v in leaves.items()})
# cross-check of the WRITER in track_mjx_amd/h5lite.py: a file it writes must read back identically through the real h5py
import sys, tempfile
sys.path.insert(0, str(HERE.parents[1]))
from track_mjx_amd import h5lite  # noqa: E402
with tempfile.TemporaryDirectory() as d:
    out = Path(d) / "written_by_h5lite.h5"
    payload = {"qpos": qpos, "qvel": qvel, "xpos": xpos, "xquat": xquat, "config": config.encode("utf-8"), "offsets": np.arange(12, dtype=np.int64).reshape(3, 4)}
    h5lite.write_file(out, payload)
    with h5py.File(out, "r") as f:
        assert sorted(f.keys()) == sorted(payload)
        for k, v in payload.items():
            got = f[k][()]
            assert (got == v) if isinstance(v, bytes) else (got.dtype == v.dtype and np.array_equal(got, v)), k
print("h5lite.write_file -> h5py round trip ok")
print("wrote", [p.name for p in HERE.glob("*.h5")])
