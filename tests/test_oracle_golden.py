"""CPU: the C oracle against the committed golden vectors (tests/golden/task_golden.npz, made by make_golden.py)."""
from pathlib import Path

import numpy as np
import pytest

from tests.common import default_blob, default_walker, make_oracle
from track_mjx_amd import clips as _clips

G = np.load(Path(__file__).parent / "golden" / "task_golden.npz")
# the reference's other shipped rodent configuration (rodent-sps-per-actor.yaml: CG 4 / 4, 5 substeps per control step, penalty scale
# [1, 1, 0.2], RewardConfig's default var / jerk coefficients): step cases + its own frame-index table (0.01 s per control step)
G_SPS = np.load(Path(__file__).parent / "golden" / "task_golden_sps.npz")


@pytest.fixture(scope="module")
def oracle():
    w, cfg = default_walker()
    clip = _clips.make_synthetic_clips(w.model, 3, seed=123)
    return make_oracle(default_blob(w, cfg), clip, "f32"), w


@pytest.fixture(scope="module")
def oracle_sps():
    w, cfg = default_walker("rodent-sps-per-actor")
    clip = _clips.make_synthetic_clips(w.model, 3, seed=123)
    return make_oracle(default_blob(w, cfg), clip, "f32"), w


def _run_case(O, envs, i, G=G):
    z = np.zeros(74)
    O.env_reset(envs, 0, int(G["in_clip_idx"][i]), int(G["in_start_frame"][i]), z, z[:73])
    for k in ("qpos", "qvel", "xpos", "qfrc_actuator"):
        O.env_set(envs, 0, k, G["in_" + k][i])
    O.env_set(envs, 0, "time", [G["in_time"][i]])
    xm = np.zeros(68 * 9); xm[27:36] = G["in_xmat_torso"][i]
    O.env_set(envs, 0, "xmat", xm)
    O.env_set(envs, 0, "action_buffer", G["in_action_buffer"][i]); O.env_set(envs, 0, "buffer_index", [G["in_buffer_index"][i]])
    O.env_post(envs, 0, G["in_action"][i])


@pytest.mark.parametrize("which", ["rodent-full-clips", "rodent-sps-per-actor"])
def test_step_terms_match_golden(which, request):
    O, w = request.getfixturevalue("oracle" if which == "rodent-full-clips" else "oracle_sps")
    G = globals()["G" if which == "rodent-full-clips" else "G_SPS"]
    envs = O.new_envs(1)
    for i in range(G["in_qpos"].shape[0]):
        _run_case(O, envs, i, G)
        assert int(O.env_get(envs, 0, "cur_frame")[0]) == int(G["out_frame"][i]) or G["out_done"][i] > 0
        m = O.env_get(envs, 0, "metrics")
        np.testing.assert_allclose(m, G["out_metrics"][i], rtol=2e-5, atol=2e-6, err_msg=f"case {i}")
        np.testing.assert_allclose(O.env_get(envs, 0, "reward")[0], G["out_reward"][i], rtol=2e-5, atol=2e-6)
        assert O.env_get(envs, 0, "done")[0] == G["out_done"][i]
        assert int(O.env_get(envs, 0, "buffer_index")[0]) == int(G["out_buffer_index"][i])
        assert np.array_equal(O.env_get(envs, 0, "action_buffer").astype(np.float32), G["out_action_buffer"][i])
        if G["out_done"][i] == 0:   # done envs return the auto-reset snapshot instead of the step's observation
            np.testing.assert_allclose(O.env_get(envs, 0, "obs"), G["out_obs"][i], rtol=2e-5, atol=2e-6, err_msg=f"case {i}")
    assert (G["out_done"] == 0).sum() >= 3 and (G["out_done"] > 0).sum() >= 3


@pytest.mark.parametrize("which", ["rodent-full-clips", "rodent-sps-per-actor"])
def test_frame_index_table_bit_exact(which, request):
    O, w = request.getfixturevalue("oracle" if which == "rodent-full-clips" else "oracle_sps")
    G = globals()["G" if which == "rodent-full-clips" else "G_SPS"]
    envs = O.new_envs(1)
    z = np.zeros(74)
    for start in range(44):
        O.env_reset(envs, 0, 0, start, z, z[:73])
        for s in range(195):
            O.env_set(envs, 0, "time", [G["frame_times"][s]])
            assert int(O.env_get(envs, 0, "cur_frame")[0]) == int(G["frame_table"][s, start])
    if which == "rodent-full-clips":
        # documented example: start 0, steps 4..8 land on frames 3..7 (t = 0.07999998 at step 4)
        assert list(G["frame_table"][3:8, 0]) == [3, 4, 5, 6, 7]
    else:
        assert G["frame_table"][-1, 0] in (96, 97)      # 195 control steps of 0.01 s at 50 Hz


def test_gae_matches_golden_and_closed_form(oracle):
    O, _ = oracle
    vs, adv = O.gae(G["gae_trunc"], G["gae_term"], G["gae_rew"], G["gae_val"], G["gae_boot"], 0.95, 0.98)
    np.testing.assert_allclose(vs, G["gae_vs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(adv, G["gae_adv"], rtol=1e-5, atol=1e-6)
    # third formulation: explicit double sum  vs_t - V_t = sum_k (prod_{j<k} c_{t+j}) delta_{t+k}
    T, B = G["gae_rew"].shape
    tm = 1 - G["gae_trunc"].astype(np.float64); term = G["gae_term"].astype(np.float64); val = G["gae_val"].astype(np.float64)
    v1 = np.concatenate([val[1:], G["gae_boot"][None].astype(np.float64)], 0)
    delta = (G["gae_rew"] + 0.98 * (1 - term) * v1 - val) * tm
    c = 0.98 * (1 - term) * tm * 0.95
    ref = np.zeros((T, B))
    for t in range(T):
        w = np.ones(B)
        for k in range(t, T):
            ref[t] += w * delta[k]
            w = w * c[k]
    np.testing.assert_allclose(vs - val, ref, rtol=1e-4, atol=1e-5)


def test_bounded_quat_dist_known_answers():
    exp = G["bqd_expect"]
    assert abs(exp[0]) < 1e-3 and abs(exp[1] - np.pi / 4) < 1e-6 and abs(exp[2]) < 1e-3 and abs(exp[3] - np.pi / 2) < 1e-6
