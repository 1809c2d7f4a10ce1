"""CPU: the HIP kernel bodies (tests/lane/physics_core.h, csrc/env_core.h) compiled for the host by the TEST-ONLY emulation
harness tests/hostemu/ and compared with the oracle.  This is how the kernel source is unit-tested in the GPU-less build
container; the same checks run on the real GPU in test_gpu_parity.py.  Nothing in the product loads the emulation."""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).parent / "hostemu"))
from emu import Emu  # noqa: E402

from tests.common import default_blob, default_walker, make_oracle, rel_err  # noqa: E402
from track_mjx_amd import clips as _clips  # noqa: E402

G = np.load(Path(__file__).parent / "golden" / "task_golden.npz")


@pytest.fixture(scope="module")
def setup():
    w, cfg = default_walker()
    blob = default_blob(w, cfg)
    clip = _clips.make_synthetic_clips(w.model, 4, seed=0)
    return w, blob, clip


def _states(clip, n, rng, sink):
    qpos = np.zeros((n, 74)); qvel = rng.uniform(-1e-3, 1e-3, size=(n, 73))
    for e in range(n):
        c, f = e % 4, (7 * e) % 44
        qpos[e] = np.concatenate([clip.position[c, f], clip.quaternion[c, f], clip.joints[c, f]]) + rng.uniform(-1e-3, 1e-3, 74)
        qpos[e, 2] -= sink * (e % 5)
    return qpos, qvel


@pytest.mark.parametrize("impl", ["lane", "wave"])
def test_forward_intermediates_sparse_vs_dense(setup, impl):
    """impl = lane: tests/lane/physics_core.h (lane-per-env reference kernel body); impl = wave: csrc/wave_physics.h (the product
    kernel: 64 emulated lanes + an LDS image per env)."""
    w, blob, clip = setup
    n = 8
    E = Emu(blob, n); O64 = make_oracle(blob, clip, "f64")
    rng = np.random.default_rng(0)
    qpos, qvel = _states(clip, n, rng, 0.012)
    act = rng.uniform(-0.1, 0.1, size=(n, 38))
    E.rows("qpos")[:] = qpos.T; E.rows("qvel")[:] = qvel.T; E.rows("act")[:] = act.T
    (E.physics if impl == "lane" else E.physics_wave)(None, 1, do_euler=False)
    ds = []
    for e in range(n):
        d = O64.new_data(qpos[e], qvel[e]); O64.set(d, "act", act[e]); O64.forward(d); ds.append(d)
    names = (("xpos", 2e-6), ("cdof", 5e-6), ("qfrc_smooth", 2e-5), ("con_dist", 5e-6), ("con_frame", 5e-6),
                      ("efc_D", 5e-5), ("efc_aref", 2e-5), ("qacc_smooth", 2e-4), ("qacc", 2e-4), ("efc_force", 2e-4))
    for name, tol in names:
        ref = np.stack([O64.get(d, name) for d in ds], 1)
        assert rel_err(E.rows(name), ref) < tol, (name, rel_err(E.rows(name), ref))
    assert (E.rows("con_dist") < 0).sum() > 0
    assert np.array_equal(E.rows("con_dist") < 0, np.stack([O64.get(d, "con_dist") < 0 for d in ds], 1))


@pytest.mark.parametrize("impl", ["lane", "wave", "wave-generic", "wave-cg4"])
def test_substeps_teacher_forced(setup, impl, monkeypatch):
    """wave: the rodent's register-resident chain path (what the static GPU kernel runs); wave-generic: TMJX_EMU_GENERIC=1, the
    LDS-resident sparse path any other tree takes; wave-cg4: the chain path with the solver options of the reference's other shipped
    rodent configuration (rodent-sps-per-actor.yaml:13-14: CG 4 / 4 — run-time constants of the blob, not kernel constants)."""
    if impl == "wave-generic":
        monkeypatch.setenv("TMJX_EMU_GENERIC", "1")
    w, blob, clip = setup
    if impl == "wave-cg4":
        w4, cfg4 = default_walker("rodent-sps-per-actor")
        blob = default_blob(w4, cfg4)
        impl = "wave"
    n = 8
    E = Emu(blob, n); O32 = make_oracle(blob, clip, "f32"); O64 = make_oracle(blob, clip, "f64")
    rng = np.random.default_rng(1)
    qpos, qvel = _states(clip, n, rng, 0.001)
    d32 = [O32.new_data(qpos[e], qvel[e]) for e in range(n)]; d64 = [O64.new_data(qpos[e], qvel[e]) for e in range(n)]
    names = ("qpos", "qvel", "act", "qacc_warmstart", "time")
    acc = {k: ([], []) for k in ("qpos", "qvel")}
    for sub in range(30):
        a = np.clip(rng.normal(size=(n, 38)) * 0.03, -1, 1)
        st = {k: np.stack([O64.get(d, k) for d in d64], 1) for k in names}
        for k, v in st.items():
            E.rows(k)[:] = v
            for e in range(n):
                O32.set(d32[e], k, v[:, e])
        if impl == "lane":
            E.physics(a.T.astype(np.float32).copy(), 1, True)
        else:
            E.physics_wave(a.T.astype(np.float32).copy(), 1, True, dump=False)
        for e in range(n):
            O32.step(d32[e], a[e]); O64.step(d64[e], a[e])
        for k in ("qpos", "qvel"):
            ref = np.stack([O64.get(d, k) for d in d64], 1); r32 = np.stack([O32.get(d, k) for d in d32], 1)
            e_emu, e_32 = rel_err(E.rows(k), ref, axis=0), rel_err(r32, ref, axis=0)
            acc[k][0].append(e_emu); acc[k][1].append(e_32)
            assert np.median(e_emu) <= 1e-5, (sub, k, e_emu)
    for k in ("qpos", "qvel"):
        g, f = np.concatenate(acc[k][0]), np.concatenate(acc[k][1])
        # over the whole run (240 env-substeps): the chain path — trunk block of the factorisation in float64, wave_physics.h — is at least as
        # accurate as the float32 restatement of MJX's dense path on the typical env-substep; the generic / lane paths (plain float32 leaf -> root
        # L^T D L) sit ~3 x above it.  A single env-substep can land an order of magnitude off either way (five non-converged CG iterations
        # amplify rounding differences), hence quantiles over the run and a loose bound on the worst one
        assert np.median(g) <= (1.25 if impl == "wave" else 4.0) * np.median(f) + 1e-7, (k, np.median(g), np.median(f))
        assert np.quantile(g, 0.9) <= (2.0 if impl == "wave" else 4.0) * np.quantile(f, 0.9) + 1e-5, (k, np.quantile(g, 0.9), np.quantile(f, 0.9))
        assert g.max() <= 4 * f.max() + (1e-4 if k == "qvel" else 2e-5), (k, g.max(), f.max())
    assert sum((O64.get(d, "con_dist") < 0).sum() for d in d64) > 0


def test_env_step_and_golden_post(setup):
    w, blob, clip = setup
    n = 4
    E = Emu(blob, n); O = make_oracle(blob, clip, "f32")
    E.set_clips(clip.as_dict())
    rng = np.random.default_rng(3)
    ci = (np.arange(n) % 4).astype(np.int32); sf = ((np.arange(n) * 11) % 44).astype(np.int32)
    qn = rng.uniform(-1e-3, 1e-3, (74, n)).astype(np.float32); vn = rng.uniform(-1e-3, 1e-3, (73, n)).astype(np.float32)
    E.reset(ci, sf, qn, vn)
    envs = O.new_envs(n)
    for e in range(n):
        O.env_reset(envs, e, ci[e], sf[e], qn[:, e], vn[:, e])
    assert np.abs(E.obs - np.stack([O.env_get(envs, e, "obs") for e in range(n)], 1)).max() < 1e-6
    for s in range(2):
        a = np.clip(rng.normal(size=(38, n)) * 0.03, -1, 1).astype(np.float32)
        E.step(a)
        for e in range(n):
            O.env_step(envs, e, a[:, e])
        assert rel_err(E.obs, np.stack([O.env_get(envs, e, "obs") for e in range(n)], 1)) < 1e-4
        assert np.abs(E.reward - np.array([O.env_get(envs, e, "reward")[0] for e in range(n)])).max() < 1e-5
    # K3 alone on the golden cases (clips of the golden file: seed 123, 3 clips)
    gclip = _clips.make_synthetic_clips(w.model, 3, seed=123)
    m = G["in_qpos"].shape[0]
    E2 = Emu(blob, m); E2.set_clips(gclip.as_dict())
    z = np.zeros((74, m), np.float32)
    E2.reset(G["in_clip_idx"].astype(np.int32), G["in_start_frame"].astype(np.int32), z, z[:73])
    first_obs = E2.obs.copy()
    for k in ("qpos", "qvel", "xpos", "xmat_torso", "qfrc_actuator"):
        E2.rows(k)[:] = G["in_" + k].T
    E2.rows("time")[:] = G["in_time"][None]
    lay_buf = E2.st  # action buffer rows follow prev_ctrl: locate through a write/read of the kernel itself
    E2.ist[2] = G["in_buffer_index"]
    # action_buffer offset = rows("qfrc_actuator") end + nu (layout: ..., qfrc_actuator, prev_ctrl, action_buffer)
    off = 259 + 204 + 9 + 73 + 38
    E2.st[off:off + 1900] = G["in_action_buffer"].T
    E2.post(G["in_action"].T.astype(np.float32).copy())
    np.testing.assert_allclose(E2.metrics.T, G["out_metrics"], rtol=3e-5, atol=3e-6)
    np.testing.assert_allclose(E2.reward, G["out_reward"], rtol=3e-5, atol=3e-6)
    assert np.array_equal(E2.done, G["out_done"]) and np.array_equal(E2.ist[2], G["out_buffer_index"])
    keep = G["out_done"] == 0
    np.testing.assert_allclose(E2.obs.T[keep], G["out_obs"][keep], rtol=3e-5, atol=3e-6)
    assert np.array_equal(E2.obs.T[~keep], first_obs.T[~keep])   # auto-reset returns the snapshot bit-exactly


def test_many_active_rows_multi_slot(setup):
    """More than 64 active constraint rows (every joint pushed past a limit, the body sunk into the floor): exercises the multi-slot
    path of the compacted constraint rows (wave_physics.h: tmw_make_constraint) against the float64 oracle, forward intermediates."""
    w, blob, clip = setup
    n = 4
    E = Emu(blob, n); O64 = make_oracle(blob, clip, "f64")
    rng = np.random.default_rng(5)
    qpos, qvel = _states(clip, n, rng, 0.0)
    rngs = np.asarray(w.model["jnt_range"]).reshape(-1, 2)[1:]          # hinge joints (joint 0 is the free joint)
    for e in range(n):
        over = rng.uniform(0.005, 0.03, size=67) * (rngs[:, 1] - rngs[:, 0])
        qpos[e, 7:] = np.where(rng.random(67) < 0.5, rngs[:, 1] + over, rngs[:, 0] - over)
        qpos[e, 2] -= 0.03 * (e + 1)
    act = rng.uniform(-0.1, 0.1, size=(n, 38))
    E.rows("qpos")[:] = qpos.T; E.rows("qvel")[:] = qvel.T; E.rows("act")[:] = act.T
    E.physics_wave(None, 1, do_euler=False)
    ds = []
    for e in range(n):
        d = O64.new_data(qpos[e], qvel[e]); O64.set(d, "act", act[e]); O64.forward(d); ds.append(d)
    # by construction all 67 limits are violated (+ 4 rows per penetrating contact): more than one 64-row slot of active rows
    for name, tol in (("efc_D", 5e-5), ("efc_aref", 5e-5), ("qacc_smooth", 5e-4), ("efc_force", 5e-3), ("qacc", 5e-3)):
        ref = np.stack([O64.get(d, name) for d in ds], 1)
        assert rel_err(E.rows(name), ref) < tol, (name, rel_err(E.rows(name), ref))


@pytest.mark.parametrize("R", [2, 3])
def test_action_repeat_of_the_episode_wrapper(setup, R):
    """wrappers.wrap(env, action_repeat=R) (wrappers.py:21,43 -> brax EpisodeWrapper.step): the kernel's K3 flags (env_core.h: TM_REP_*) against
    (1) R single steps of the same functions WITHOUT wrappers' resets in between (a twin restarted from the same buffers every call: an env
    that terminates in an early repeat keeps stepping, as in brax's scan) and (2) the oracle's EpisodeWrapper restatement; episode_length 7
    -> truncation once the counter, advancing by R, reaches it; the counter restarts after a done."""
    _, _, clip = setup
    n, ep = 4, 7
    blob = default_blob(episode_length=ep)
    E = Emu(blob, n); E1 = Emu(default_blob(episode_length=1000, auto_reset=False), n); O = make_oracle(blob, clip, "f32")
    O.set_action_repeat(R)
    E.set_clips(clip.as_dict()); E1.set_clips(clip.as_dict())
    rng = np.random.default_rng(5)
    ci = (np.arange(n) % 4).astype(np.int32); sf = ((np.arange(n) * 7) % 30).astype(np.int32)
    qn = rng.uniform(-1e-3, 1e-3, (74, n)).astype(np.float32); vn = rng.uniform(-1e-3, 1e-3, (73, n)).astype(np.float32)
    qn[2] += 0.05       # start 5 cm up: the first call's R env steps are contact-free, so the values can be held against the oracle tightly
    E.reset(ci, sf, qn, vn)
    first_obs = E.obs.copy()
    envs = O.new_envs(n)
    for e in range(n):
        O.env_reset(envs, e, ci[e], sf[e], qn[:, e], vn[:, e])
    steps, prev_done, seen_trunc, seen_term, seen_restart = np.zeros(n, np.float32), np.zeros(n, bool), False, False, False
    for call in range(10):
        a = np.clip(rng.normal(size=(38, n)) * 0.03, -1, 1).astype(np.float32)
        E1.st[:] = E.st; E1.ist[:] = E.ist
        total = np.zeros(n, np.float32)
        for r in range(R):
            E1.step(a); total = total + E1.reward
        inner_done = E1.done > 0
        E.step_repeat(a, R)
        seen_restart |= bool((prev_done & (steps > 0)).any())
        steps = np.where(prev_done, 0, steps) + R
        over = steps >= ep
        done = inner_done | over
        assert np.array_equal(E.rows("steps")[0], steps)
        assert np.array_equal(E.done > 0, done) and np.array_equal(E.trunc > 0, over & ~inner_done)
        assert np.array_equal(E.reward, total) and np.array_equal(E.metrics, E1.metrics)
        assert np.array_equal(E.obs[:, ~done], E1.obs[:, ~done]) and np.array_equal(E.obs[:, done], first_obs[:, done])
        assert np.array_equal(E.rows("qpos")[:, ~done], E1.rows("qpos")[:, ~done]) and np.array_equal(E.rows("qvel")[:, ~done], E1.rows("qvel")[:, ~done])
        seen_trunc |= bool((over & ~inner_done).any()); seen_term |= bool(inner_done.any())
        prev_done = done
        if call == 0:
            # against the oracle on the first call only: once the feet load up, the synthetic poses' contacts make free-running float32
            # trajectories part ways within one env step (single steps too)
            for e in range(n):
                O.env_step(envs, e, a[:, e])
            assert np.array_equal(np.array([O.env_get(envs, e, "steps")[0] for e in range(n)]), steps)
            assert np.array_equal(np.array([O.env_get(envs, e, "done")[0] for e in range(n)]) > 0, done)
            rew_err = np.abs(E.reward - np.array([O.env_get(envs, e, "reward")[0] for e in range(n)])).max()
            obs_err = rel_err(E.obs, np.stack([O.env_get(envs, e, "obs") for e in range(n)], 1))
            print(f"action_repeat {R}: summed reward err {rew_err:.2e}, obs rel err {obs_err:.2e} against the float32 oracle")
            assert rew_err < 2e-5 * R and obs_err < 2e-4
    assert seen_trunc and seen_restart, (seen_trunc, seen_term, seen_restart)


def test_wave_kernel_results_do_not_depend_on_what_a_wave_finds_when_it_starts(setup, monkeypatch):
    """A wave starts on whatever the previous wave of its CU left in LDS, in the registers and in its scratch — NaNs, if that env had blown up.
    TMJX_EMU_POISON=nan starts the emulated wave from NaNs instead of zeros: free-running through contact, every physics row and output must come
    out bit for bit as from the zero image.  (Round 5: the branch-free row products cancel out-of-range words by a zero factor; with cinert moved
    onto xipos some of those words were no longer rewritten every substep, and the GPU's `action_repeat` twin test turned flaky under full-scale
    actions — tmw_load_state now zero-fills the env's LDS image.)"""
    w, blob, clip = setup
    n = 12
    res = []
    for poison in (False, True):
        if poison:
            monkeypatch.setenv("TMJX_EMU_POISON", "nan")
        else:
            monkeypatch.delenv("TMJX_EMU_POISON", raising=False)
        E = Emu(blob, n)
        rng = np.random.default_rng(11)
        qpos, qvel = _states(clip, n, rng, 0.001)
        E.rows("qpos")[:] = qpos.T; E.rows("qvel")[:] = qvel.T
        E.physics_wave(None, 1, do_euler=False)
        out, contact = [], 0
        for ctl in range(2):
            a = np.clip(rng.normal(size=(n, 38)) * (0.3 if ctl == 0 else 1.0), -1, 1).T.astype(np.float32).copy()
            for sub in range(10):
                E.physics_wave(a, 1, True, dump=True)
                out.append({k: E.rows(k).copy() for k in ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "qfrc_actuator", "solver_stats")})
                contact += int((E.rows("con_dist") < 0).any(0).sum())
        res.append(out)
        assert contact > 20
    for t, (x, y) in enumerate(zip(*res)):
        for k in x:
            same = (x[k] == y[k]) | (np.isnan(x[k]) & np.isnan(y[k]))
            assert same.all(), (t, k, int((~same).sum()))
