"""The numpy HDF5 reader (track_mjx_amd/h5lite.py) and the clip loaders on top of it (track_mjx_amd/io/load.py, mirror of
track_mjx/io/load.py) against files written by the REAL h5py 3.3.0 / libhdf5 1.10.6 (tests/golden/make_h5_fixtures.py)."""
from pathlib import Path

import numpy as np
import pytest

from track_mjx_amd import h5lite
from track_mjx_amd.io import load

G = Path(__file__).resolve().parent / "golden"
E = np.load(G / "h5_expected.npz")


@pytest.mark.parametrize("name", ["stac_small.h5", "stac_latest.h5"])
def test_datasets_bit_exact(name):
    with h5lite.File(G / name) as f:
        assert sorted(f.keys()) == ["config", "kp_names", "offsets", "qpos", "qvel", "xpos", "xquat"]
        for k in ("qpos", "qvel", "xpos", "xquat"):     # contiguous f32, contiguous f64, chunked gzip+shuffle, chunked raw
            a = f[k][()]
            assert a.dtype == E[k].dtype and a.shape == E[k].shape and np.array_equal(a, E[k]), k
            assert f[k].shape == E[k].shape
        assert f["config"][()] == E["config"].tobytes()                       # variable-length UTF-8 string (global heap)
        assert f["kp_names"][()].tolist() == [b"Snout", b"EarL", b"EarR"]      # fixed-length strings
        assert np.array_equal(f["offsets"][()], np.arange(12).reshape(3, 4)) and f["offsets"].dtype == np.int64
        assert "qpos" in f and "nope" not in f
        with pytest.raises(KeyError):
            f["nope"]


def test_groups_nested_and_large():
    with h5lite.File(G / "refclip_small.h5") as f:
        assert sorted(f.keys()) == ["all_clips", "config", "many", "meta"]
        assert float(f["meta/inner/scalar"][()]) == 2.5 and float(f["meta"]["inner"]["scalar"][()]) == 2.5
        many = f["many"]                                                        # 270 links: two B-tree levels
        assert len(many.keys()) == 270 and all(int(many[f"d{i:03d}"][()]) == i for i in (0, 7, 8, 131, 269))
        assert f["config"][()].startswith(b"stac:")


def test_not_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(h5lite.H5Error):
        h5lite.File(p)


def test_load_data_stac_layout():
    clip = load.load_data(str(G / "stac_small.h5"))        # n_frames_per_clip from the config YAML
    assert clip.position.shape == (3, 10, 3) and clip.body_positions.shape == (3, 10, 67, 3) and clip.joints.shape == (3, 10, 67)
    q = E["qpos"].reshape(3, 10, 74)
    assert np.array_equal(clip.quaternion, q[:, :, 3:7]) and np.array_equal(clip.joints, q[:, :, 7:])
    assert np.array_equal(clip.joints_velocity, E["qvel"].reshape(3, 10, 73)[:, :, 6:].astype(np.float32))
    assert clip.position.dtype == np.float32
    assert load.make_multiclip_data(str(G / "stac_small.h5"), n_frames_per_clip=15).position.shape == (2, 15, 3)
    assert load.make_singleclip_data(str(G / "stac_small.h5"))[0].position.shape == (30, 3)
    assert load.load_clips_metadata(str(G / "stac_small.h5")) == [("Walk", 12), ("Rear", 3), ("FaceGroom", 7)]


def test_load_data_falls_back_to_reference_clip_layout():
    clip = load.load_data(str(G / "refclip_small.h5"))     # no qpos dataset -> KeyError -> group `all_clips`
    for k in ("position", "quaternion", "joints", "body_positions", "velocity", "angular_velocity", "joints_velocity", "body_quaternions"):
        assert np.array_equal(getattr(clip, k), E["leaf_" + k]), k
    with pytest.raises(KeyError):
        load.load_reference_clip_data(str(G / "stac_small.h5"))
    with pytest.raises(FileNotFoundError):
        load.load_reference_clip_data(str(G / "missing.h5"))


def test_train_test_split_and_select():
    clip = load.load_data(str(G / "refclip_small.h5"))
    big = load.select_clips(clip, np.arange(30) % 3)
    np.random.seed(0)
    train, test = load.generate_train_test_split(big, test_ratio=0.2)
    assert test.position.shape[0] == 6 and train.position.shape[0] == 24
    ids = np.concatenate([train.original_clip_idx[:, 0], test.original_clip_idx[:, 0]])
    assert sorted(ids.tolist()) == list(range(30)) and train.original_clip_idx.shape == (24, 1)
    assert np.array_equal(test.joints, big.joints[test.original_clip_idx[:, 0]])
    np.random.seed(1)
    sub = load.sub_sample_training_set(train.original_clip_idx[:, 0], 0.5)
    assert len(sub) == 12 and np.all(np.diff(sub) > 0) and set(sub) <= set(train.original_clip_idx[:, 0].tolist())


def test_train_entrypoint_clip_selection(tmp_path):
    """track_mjx_amd.train.load_clip_sets = the clip selection of track_mjx/train.py:163-209 (file, JSON split, random split)."""
    import json
    from track_mjx_amd import config as C
    from track_mjx_amd.train import load_clip_sets
    cfg = C.default_config()
    cfg["data_path"] = str(G / "refclip_small.h5")
    cfg["train_setup"]["train_subset_ratio"] = None
    train, test = load_clip_sets(cfg)
    assert train.position.shape == (3, 10, 3) and test is None
    split = tmp_path / "split.json"
    split.write_text(json.dumps({"train": [0, 2], "test": [1], "train_subset": {"0.50": [2]}}))
    cfg["train_setup"]["train_test_split_info"] = str(split)
    train, test = load_clip_sets(cfg)
    assert train.original_clip_idx.ravel().tolist() == [0, 2] and test.original_clip_idx.ravel().tolist() == [1]
    cfg["train_setup"]["train_subset_ratio"] = 0.5
    train, test = load_clip_sets(cfg)
    assert train.original_clip_idx.ravel().tolist() == [2] and np.array_equal(train.joints[0], E["leaf_joints"][2])


def test_writer_round_trip_and_clip_export(tmp_path):
    """h5lite.write_file (also read back by the real h5py in tests/golden/make_h5_fixtures.py) and tools/make_clips_h5.export:
    synthetic clips -> stac-mjx file -> io.load.load_data gives the clips back bit for bit."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import make_clips_h5
    from tests.common import default_walker
    from track_mjx_amd import clips as _clips
    payload = {"a": np.arange(6, dtype=np.int32).reshape(2, 3), "b": np.linspace(0, 1, 5), "c": b"hello: world\n"}
    h5lite.write_file(tmp_path / "w.h5", payload)
    with h5lite.File(tmp_path / "w.h5") as f:
        assert sorted(f.keys()) == ["a", "b", "c"] and f["c"][()] == payload["c"]
        assert np.array_equal(f["a"][()], payload["a"]) and f["a"].dtype == np.int32 and np.array_equal(f["b"][()], payload["b"])
    with pytest.raises(h5lite.H5Error):
        h5lite.write_file(tmp_path / "x.h5", {f"d{i}": np.zeros(1) for i in range(9)})
    w, _ = default_walker()
    cl = _clips.make_synthetic_clips(w.model, 3, n_frames=40)
    make_clips_h5.export(cl, tmp_path / "clips.h5")
    back = load.load_data(str(tmp_path / "clips.h5"))
    for k in ("position", "quaternion", "joints", "body_positions", "velocity", "angular_velocity", "joints_velocity", "body_quaternions"):
        assert np.array_equal(getattr(back, k), getattr(cl, k)), k
    assert load.load_clips_metadata(str(tmp_path / "clips.h5")) == [("Synth", 0), ("Synth", 1), ("Synth", 2)]
