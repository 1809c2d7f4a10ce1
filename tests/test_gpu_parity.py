"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Tolerances (fp32 path, BASELINE.json asks for 1e-5 rel on qpos/qvel):
  * one physics substep from an identical state:  median rel err <= 1e-5 over envs, and the HIP error against
    the float64 oracle must stay within 4x of the float32 oracle's own error against float64 (+1e-5): contact
    dynamics with a 5-iteration CG is chaotic, so trajectories are compared teacher-forced, not free-running
    (DESIGN.md "Parity method").
  * integer paths (frame index, contact slot -> geom pair, active sets, done flags): bit exact.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.common import default_blob, default_walker, make_env_and_oracle, make_oracle, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_states(O, datas, names=("qpos", "qvel", "act", "qacc_warmstart", "time")):
    return {k: np.stack([O.get(d, k) for d in datas], 1) for k in names}


def _push(env, st):
    for k, v in st.items():
        env.rows(k).copy_(torch.from_numpy(v.astype(np.float32)))


def _init_states(cl, n, rng, sink=0.0):
    qpos = np.zeros((n, 74)); qvel = rng.uniform(-1e-3, 1e-3, size=(n, 73))
    for e in range(n):
        c, f = e % cl.position.shape[0], (7 * e) % 44
        qpos[e] = np.concatenate([cl.position[c, f], cl.quaternion[c, f], cl.joints[c, f]]) + rng.uniform(-1e-3, 1e-3, 74)
        qpos[e, 2] -= sink * (e % 5)
    return qpos, qvel


def test_product_handle_launches_the_static_rodent_kernel():
    """The rodent model must take the compile-time specialisation of the physics kernel (register-resident chain kernels, lean LDS map:
    at most 9 LDS granules of 1280 bytes = 14 envs per CU).  model_host.h falls back to the generic kernel when the loaded model does not match its compile-time tables — silently,
    and every parity test still passes on that path at 0.4 x the speed (it happened once, through an over-strict host check)."""
    import ctypes as C
    env, _, _ = make_env_and_oracle(num_envs=8, wrappers=True)
    r0, cnt = C.c_int32(), C.c_int32()
    assert env._L.tmjx_debug_rows(env._handle, b"k2_kernel", C.byref(r0), C.byref(cnt)) == 0
    assert r0.value == 1, "the rodent handle fell back to the generic physics kernel"
    assert cnt.value <= 9 * 1280, cnt.value


@pytest.mark.gpu
def test_forward_intermediates():
    env, O, cl = make_env_and_oracle(num_envs=32, wrappers=False)
    O64 = make_oracle(env._blob, cl, "f64")
    rng = np.random.default_rng(0)
    qpos, qvel = _init_states(cl, 32, rng, sink=0.003)
    act = rng.uniform(-0.1, 0.1, size=(32, 38))
    env.rows("qpos").copy_(torch.from_numpy(qpos.T.astype(np.float32)))
    env.rows("qvel").copy_(torch.from_numpy(qvel.T.astype(np.float32)))
    env.rows("act").copy_(torch.from_numpy(act.T.astype(np.float32)))
    env.forward()
    torch.cuda.synchronize()
    ds = []
    for e in range(32):
        d = O64.new_data(qpos[e], qvel[e]); O64.set(d, "act", act[e]); O64.forward(d); ds.append(d)
    par = env.walker.model["dof_parentid"]

    def sparse(dense):
        out = []
        for i in range(73):
            j = i
            while j >= 0:
                out.append(dense[i * 73 + j]); j = par[j]
        return np.array(out)
    checks = {"xpos": None, "cdof": None, "qfrc_actuator": None, "qfrc_smooth": None, "con_dist": None,
              "con_frame": None, "efc_D": None, "efc_aref": None, "qacc_smooth": None, "qacc": None, "efc_force": None}
    for name in checks:
        got = env.rows(name).cpu().numpy().astype(np.float64)
        ref = np.stack([O64.get(d, name) for d in ds], 1)
        tol = 2e-4 if name in ("qacc", "efc_force", "qacc_smooth") else 2e-5
        assert rel_err(got, ref) < tol, (name, rel_err(got, ref))
    gotM = env.rows("qM").cpu().numpy().astype(np.float64)
    refM = np.stack([sparse(O64.get(d, "qM")) for d in ds], 1)
    assert rel_err(gotM, refM) < 2e-6
    # contact active sets and limit active sets are bit-exact integer paths
    act_gpu = (env.rows("con_dist").cpu().numpy() < 0)
    act_ref = np.stack([O64.get(d, "con_dist") < 0 for d in ds], 1)
    assert (act_gpu == act_ref).all()
    assert act_ref.sum() > 0, "test state must exercise contacts"


@pytest.mark.parametrize("variant,config", [("product", "rodent-full-clips"), ("product", "rodent-sps-per-actor"), ("generic-tree", "rodent-full-clips"),
                                            ("generic-tree", "rodent-sps-per-actor"), ("lane-per-env", "rodent-full-clips")])
def test_substep_teacher_forced(variant, config, monkeypatch):
    """product: the static rodent kernel (register-resident chain path); generic-tree: TMJX_WAVE_DYNAMIC=1, the run-time-layout
    wave kernel any other model gets (LDS-resident sparse factorisation); lane-per-env: TMJX_IMPL=lane (tests/lane/physics_core.h).
    config: the reference's two shipped rodent configurations — rodent-full-clips.yaml:12-14 (CG 5 / 5) and rodent-sps-per-actor.yaml:13-15
    (CG 4 / 4 = the constructor defaults of multi_clip_tracking.py:24-25): the solver's iteration counts are run-time constants of the handle."""
    if variant == "generic-tree":
        monkeypatch.setenv("TMJX_WAVE_DYNAMIC", "1")
    elif variant == "lane-per-env":
        # the cross-check build (-DTMJX_LANE_IMPL, built next to the tests by __graft_entry__.build(); NOT the product library)
        from track_mjx_amd import hip
        lane_so = Path(__file__).parent / "lane" / "libtmjx_hip_lane.so"
        if not lane_so.exists():
            pytest.skip("tests/lane/libtmjx_hip_lane.so not built")
        monkeypatch.setattr(hip, "_lib", hip.load(lane_so))
        monkeypatch.setenv("TMJX_IMPL", "lane")
    n = 64
    env, O32, cl = make_env_and_oracle(num_envs=n, wrappers=False, config=config)
    assert env._opts["iterations"] == (5 if config == "rodent-full-clips" else 4)
    O64 = make_oracle(env._blob, cl, "f64")
    rng = np.random.default_rng(1)
    qpos, qvel = _init_states(cl, n, rng, sink=0.001)
    d32 = [O32.new_data(qpos[e], qvel[e]) for e in range(n)]
    d64 = [O64.new_data(qpos[e], qvel[e]) for e in range(n)]
    worst = []
    for sub in range(40 if variant == "product" else 12):
        a = np.clip(rng.normal(size=(n, 38)) * 0.03, -1, 1)
        st = _oracle_states(O64, d64)
        _push(env, st)
        for e in range(n):   # float32 oracle restarts from the float64 state too
            for k, v in st.items():
                O32.set(d32[e], k, v[:, e])
        env.physics(torch.from_numpy(a.T.astype(np.float32)).contiguous().to(DEV), 1)
        torch.cuda.synchronize()
        for e in range(n):
            O32.step(d32[e], a[e]); O64.step(d64[e], a[e])
        ref = _oracle_states(O64, d64, ("qpos", "qvel")); r32 = _oracle_states(O32, d32, ("qpos", "qvel"))
        for k in ("qpos", "qvel"):
            got = env.rows(k).cpu().numpy()
            e_gpu, e_32 = rel_err(got, ref[k], axis=0), rel_err(r32[k], ref[k], axis=0)
            worst.append((k, np.median(e_gpu), e_gpu.max(), e_32.max()))
            assert np.median(e_gpu) <= 1e-5, (sub, k, np.median(e_gpu))
            # (product: trunk block of the factorisation in float64, at least as accurate as the float32 oracle on the typical env; the generic-tree and
            # lane-per-env kernels keep the plain float32 leaf -> root L^T D L, ~3 x above it)
            assert np.quantile(e_gpu, 0.9) <= (2 if variant == "product" else 4) * np.quantile(e_32, 0.9) + 1e-5, (sub, k, e_gpu.max(), e_32.max())
    ncon = sum((O64.get(d, "con_dist") < 0).sum() for d in d64)
    assert ncon > 0, "trajectory must reach contact"
    print("substep parity (median, max, f32-oracle max):", max(w[1] for w in worst), max(w[2] for w in worst), max(w[3] for w in worst))


@pytest.mark.parametrize("config", ["rodent-full-clips", "rodent-sps-per-actor"])
def test_env_reset_step_and_autoreset(config):
    """config rodent-sps-per-actor: 5 substeps per control step (0.01 s of clip time per step), CG 4 / 4, penalty scale [1, 1, 0.2] and
    RewardConfig's default var / jerk coefficients (rodent-sps-per-actor.yaml:13-16,39; reward.py:51-53)."""
    n = 32
    env, O, cl = make_env_and_oracle(num_envs=n, wrappers=True, config=config)
    assert env._n_frames == (10 if config == "rodent-full-clips" else 5)
    g = torch.Generator().manual_seed(5)
    clip = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32); start = torch.randint(0, 44, (n,), generator=g, dtype=torch.int32)
    qn = (torch.rand((74, n), generator=g) * 2 - 1) * 1e-3; vn = (torch.rand((73, n), generator=g) * 2 - 1) * 1e-3
    st = env.reset(g, clip, start_frame=start, qpos_noise=qn, qvel_noise=vn)
    envs = O.new_envs(n)
    for e in range(n):
        O.env_reset(envs, e, int(clip[e]), int(start[e]), qn[:, e].numpy(), vn[:, e].numpy())
    obs_o = np.stack([O.env_get(envs, e, "obs") for e in range(n)], 0)
    assert rel_err(st.obs.cpu().numpy(), obs_o) < 1e-5
    first_obs = st.obs.clone()
    # two gentle steps (pre-contact: free-running comparison is meaningful)
    for s in range(2):
        a = (torch.randn((38, n), generator=g) * 0.03).clamp(-1, 1)
        st = env.step(st, a.to(DEV)); torch.cuda.synchronize()
        for e in range(n):
            O.env_step(envs, e, a[:, e].numpy())
        obs_o = np.stack([O.env_get(envs, e, "obs") for e in range(n)], 0)
        rew_o = np.array([O.env_get(envs, e, "reward")[0] for e in range(n)])
        met_o = np.stack([O.env_get(envs, e, "metrics") for e in range(n)], 0)
        assert rel_err(st.obs.cpu().numpy(), obs_o) < 2e-4
        assert np.abs(st.reward.cpu().numpy() - rew_o).max() < 1e-4
        assert np.abs(env.metrics_buf.t().cpu().numpy() - met_o).max() < 1e-3
        frames_o = np.array([O.env_get(envs, e, "cur_frame")[0] for e in range(n)])
        assert (env._get_cur_frame().cpu().numpy() == frames_o).all()
    # violent actions -> terminations; the auto-reset must restore exactly the snapshot
    done_any = False
    for s in range(12):
        a = torch.randn((38, n), generator=g).clamp(-1, 1)
        st = env.step(st, a.to(DEV)); torch.cuda.synchronize()
        d = st.done.cpu().numpy() > 0
        if d.any():
            done_any = True
            assert torch.equal(st.obs[torch.from_numpy(d).to(DEV)], first_obs[torch.from_numpy(d).to(DEV)])
            L = env.layout
            phys = env.state_buf[L.qpos:L.qpos + 259][:, torch.from_numpy(d).to(DEV)]
            first = env.state_buf[L.first_phys:L.first_phys + 259][:, torch.from_numpy(d).to(DEV)]
            assert torch.equal(phys, first)
            assert (env.state_buf[L.prev_ctrl:L.prev_ctrl + 38][:, torch.from_numpy(d).to(DEV)] == 0).all()
        assert torch.isfinite(st.obs).all() and torch.isfinite(st.reward).all()
    assert done_any


def test_reward_obs_alone_random_states():
    """K3 alone against the oracle on arbitrary (non-physical) states: exercises every reward/obs term."""
    n = 64
    env, O, cl = make_env_and_oracle(num_envs=n, wrappers=True)
    rng = np.random.default_rng(7)
    g = torch.Generator().manual_seed(7)
    clip_t = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32); start_t = torch.randint(0, 44, (n,), generator=g, dtype=torch.int32)
    st = env.reset(g, clip_t, start_frame=start_t, qpos_noise=torch.zeros((74, n)), qvel_noise=torch.zeros((73, n)))
    envs = O.new_envs(n)
    clip = clip_t.numpy(); start = start_t.numpy()
    L = env.layout
    vals = {"qpos": rng.normal(size=(74, n)) * 0.3, "qvel": rng.normal(size=(73, n)), "xpos": rng.normal(size=(204, n)) * 0.1,
            "qfrc_actuator": rng.normal(size=(73, n)), "time": rng.integers(0, 190, size=(1, n)) * np.float32(0.02)}
    xmat = rng.normal(size=(9, n))
    buf = rng.uniform(-1, 1, size=(50 * 38, n)); bidx = rng.integers(0, 50, size=n)
    for k, v in vals.items():
        env.rows(k).copy_(torch.from_numpy(v.astype(np.float32)))
    env.rows("xmat_torso").copy_(torch.from_numpy(xmat.astype(np.float32)))
    env.state_buf[L.action_buffer:L.action_buffer + 1900].copy_(torch.from_numpy(buf.astype(np.float32)))
    env.istate_buf[L.i_buffer_index].copy_(torch.from_numpy(bidx.astype(np.int32)))
    a = rng.uniform(-1, 1, size=(38, n)).astype(np.float32)
    zero = np.zeros(74)
    for e in range(n):
        O.env_reset(envs, e, int(clip[e]), int(start[e]), zero, zero[:73])
        for k, v in vals.items():
            O.env_set(envs, e, k, v[:, e].astype(np.float32))
        xm = np.zeros(68 * 9); xm[3 * 9:4 * 9] = xmat[:, e].astype(np.float32)
        O.env_set(envs, e, "xmat", xm)
        O.env_set(envs, e, "action_buffer", buf[:, e].astype(np.float32)); O.env_set(envs, e, "buffer_index", [bidx[e]])
        O.env_post(envs, e, a[:, e])
    st = env.reward_obs(torch.from_numpy(a).to(DEV)); torch.cuda.synchronize()
    done_o = np.array([O.env_get(envs, e, "done")[0] for e in range(n)])
    keep = done_o == 0   # (done envs are overwritten by the snapshot on both sides; compare those separately)
    obs_o = np.stack([O.env_get(envs, e, "obs") for e in range(n)], 0)
    rew_o = np.array([O.env_get(envs, e, "reward")[0] for e in range(n)])
    met_o = np.stack([O.env_get(envs, e, "metrics") for e in range(n)], 0)
    assert (st.done.cpu().numpy() == done_o).all()
    assert rel_err(st.obs.cpu().numpy(), obs_o) < 1e-5
    assert np.abs(st.reward.cpu().numpy() - rew_o).max() < 1e-4 * max(1.0, np.abs(rew_o).max())
    assert rel_err(env.metrics_buf.t().cpu().numpy(), met_o) < 1e-5
    frames_o = np.array([O.env_get(envs, e, "cur_frame")[0] for e in range(n)])
    assert keep.sum() >= 0 and frames_o.min() >= 0


@pytest.mark.parametrize("config", ["rodent-full-clips", "rodent-sps-per-actor"])
def test_frame_index_table_bit_exact(config):
    """floor(time*50 + start) with time accumulated in fp32 by 10 (rodent-sps-per-actor: 5) adds of 0.002 per step: steps 1..195 x start 0..43.
    The clip's joint table is frame-coded (joints[c,f,:] = f) so the frame the KERNEL gathered is read back from the
    observation's joint rows (obs = ref_joints[frame+1+t] - qpos[7:], qpos joints = 0)."""
    from track_mjx_amd import clips as _clips
    from track_mjx_amd.environment import MultiClipTracking, RewardConfig
    n = 44
    w, cfg = default_walker(config)
    cl = _clips.make_synthetic_clips(w.model, 1, seed=0)
    cl.joints[:] = np.arange(250, dtype=np.float32)[None, :, None]
    env = MultiClipTracking(cl, w, RewardConfig(**cfg["env_config"]["reward_weights"]), **cfg["env_config"]["env_args"],
                            **cfg["reference_config"], num_envs=n, device=DEV)
    g = torch.Generator().manual_seed(0)
    env.reset(g, torch.zeros(n, dtype=torch.int32), start_frame=torch.arange(n, dtype=torch.int32))
    L = env.layout
    env.state_buf[L.qpos + 7:L.qpos + 74].zero_()
    t = np.float32(0.0); a = torch.zeros((38, n), device=DEV)
    for step in range(1, 196):
        for _ in range(cfg["env_config"]["env_args"]["physics_steps_per_control_step"]):
            t = np.float32(t + np.float32(0.002))
        env.rows("time").fill_(float(t))
        prod = np.float32(t * np.float32(50.0))
        frame = np.floor((prod + np.arange(n, dtype=np.float32)).astype(np.float32)).astype(np.int32)
        st = env.reward_obs(a)
        got = st.obs[:, 35].cpu().numpy()      # first joint row of trajectory slot t=0  -> frame + 1 (clamped)
        expect = np.clip(frame + 1, 0, 250 - 5).astype(np.float32)
        assert (got == expect).all(), (step, got, expect)
        assert (env._get_cur_frame().cpu().numpy() == frame).all()


def test_gae_matches_oracle():
    import ctypes as C
    from track_mjx_amd import hip
    O = make_oracle(default_blob())
    rng = np.random.default_rng(0)
    T, B = 20, 1024
    trunc = (rng.random((T, B)) < 0.05).astype(np.float32); term = ((rng.random((T, B)) < 0.05) * (1 - trunc)).astype(np.float32)
    rew = rng.normal(size=(T, B)).astype(np.float32); val = rng.normal(size=(T, B)).astype(np.float32); boot = rng.normal(size=B).astype(np.float32)
    vs_o, adv_o = O.gae(trunc, term, rew, val, boot, 0.95, 0.98)
    tens = [torch.from_numpy(x).to(DEV) for x in (trunc, term, rew, val, boot)]
    vs = torch.empty((T, B), device=DEV); adv = torch.empty((T, B), device=DEV)
    L = hip.lib()
    hip.check(L.tmjx_gae(*[C.c_void_p(t.data_ptr()) for t in tens], 0.95, 0.98, C.c_void_p(vs.data_ptr()), C.c_void_p(adv.data_ptr()),
                         T, B, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tmjx_gae")
    torch.cuda.synchronize()
    assert np.abs(vs.cpu().numpy() - vs_o).max() < 1e-5 and np.abs(adv.cpu().numpy() - adv_o).max() < 1e-5


def test_full_size_properties():
    """BASELINE size (4096 envs): determinism, finiteness, unit root quaternion, NaN guard -> done."""
    n = 4096
    env, O, cl = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
    outs = []
    for rep in range(2):
        g = torch.Generator().manual_seed(11)
        st = env.reset(g)
        for s in range(3):
            a = (torch.randn((38, n), generator=g) * 0.5).clamp(-1, 1).to(DEV)
            st = env.step(st, a)
        torch.cuda.synchronize()
        outs.append((st.obs.clone(), st.reward.clone(), env.state_buf.clone()))
    # bitwise run-to-run determinism (NaN-aware: an env that blew up keeps NaNs in its derived rows until its next step)
    def same(a, b):
        return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
    assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1]) and same(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    q = st.pipeline_state["qpos"][:, 3:7]
    assert (q.norm(dim=1) - 1).abs().max() < 1e-5
    # NaN injected into one env's state must raise done for that env only (single_clip_tracking.py:286-293)
    env.rows("qvel")[5, 17] = float("nan")
    st = env.step(st, torch.zeros((38, n), device=DEV)); torch.cuda.synchronize()
    assert st.done[17] == 1 and st.metrics["nan"][17] == 1
    assert torch.isfinite(st.obs).all() and torch.isfinite(st.reward).all()


def test_full_size_three_groups_against_the_oracle():
    """BASELINE size, the bench's launch layout: 4096 envs as the THREE env groups ppo.group_sizes makes (1368 / 1364 / 1364), one shared clip table.
    (1) reset + one control step (K1, 10 x K2, K3) of every group against the oracle's env for a sample of envs spread over all groups — first and
    last envs of every launch included; (2) after three more control steps that reach contact, single-substep launches of ALL envs compared
    teacher-forced with the float64 / float32 oracles on the same sample, at the strict test's kind of bounds.  A launch-size-dependent bug
    (grid tail, record stride, a third group's buffers) fails here; the oracle comparisons elsewhere run at <= 100 envs.
    Reference: track_mjx/environment/task/single_clip_tracking.py:207-320."""
    from tests.common import assert_substep_sample_bounds, oracle_substep_sample, spread_sample
    from track_mjx_amd import clips as _clips, config as _config
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.environment import wrap
    from track_mjx_amd.train import build_env
    from track_mjx_amd.walker import Rodent
    n, ncl = 4096, 64
    c = _config.default_config()
    table = _clips.make_synthetic_clips(Rodent(**c["walker_config"]).model, ncl, n_frames=c["reference_config"]["clip_length"], mocap_hz=c["env_config"]["env_args"]["mocap_hz"])
    sizes = ppo.group_sizes(n, 3)
    assert sizes == [1368, 1364, 1364]
    envs = [wrap(build_env(c, sizes[0], DEV, reference_clip=table), episode_length=195)]
    envs += [wrap(build_env(c, sz, DEV, reference_clip=table, share_clips_with=envs[0]), episode_length=195) for sz in sizes[1:]]
    O32, O64 = make_oracle(envs[0]._blob, table, "f32"), make_oracle(envs[0]._blob, table, "f64")
    g = torch.Generator().manual_seed(3)
    obs_err, rew_err, checked = [], [], 0
    states = []
    for env in envs:
        m = env.num_envs
        clip = torch.randint(0, ncl, (m,), generator=g, dtype=torch.int32); start = torch.randint(0, 44, (m,), generator=g, dtype=torch.int32)
        qn = (torch.rand((74, m), generator=g) * 2 - 1) * 1e-3; vn = (torch.rand((73, m), generator=g) * 2 - 1) * 1e-3
        st = env.reset(g, clip, start_frame=start, qpos_noise=qn, qvel_noise=vn)
        a = (torch.randn((38, m), generator=g) * 0.03).clamp(-1, 1)
        st = env.step(st, a.to(DEV))
        torch.cuda.synchronize()
        idx = spread_sample(m, 22)
        E = O32.new_envs(len(idx))
        for j, e in enumerate(idx):
            O32.env_reset(E, j, int(clip[e]), int(start[e]), qn[:, e].numpy(), vn[:, e].numpy())
            O32.env_step(E, j, a[:, e].numpy())
            o_ref = O32.env_get(E, j, "obs")
            obs_err.append(rel_err(st.obs[e].cpu().numpy(), o_ref)); rew_err.append(abs(float(st.reward[e]) - O32.env_get(E, j, "reward")[0]))
            assert float(st.done[e]) == O32.env_get(E, j, "done")[0]
            checked += 1
        states.append(st)
    assert checked >= 60 and max(obs_err) < 2e-5 and max(rew_err) < 1e-4, (max(obs_err), max(rew_err))
    for s in range(3):          # into contact, on the product path
        for k, env in enumerate(envs):
            states[k] = env.step(states[k], (torch.randn((38, env.num_envs), generator=g) * 0.5).clamp(-1, 1).to(DEV))
    torch.cuda.synchronize()
    errs = oracle_substep_sample(envs, O32, O64, np.random.default_rng(5), scale=0.3, per_group=22, substeps=3)
    print(f"\nfull size, 3 groups: first control step obs rel err max {max(obs_err):.2e}, reward abs err max {max(rew_err):.2e} on {checked} envs; teacher-forced "
          f"substeps on {len(errs['qvel'][0])} env-substeps: qvel median HIP {np.median(errs['qvel'][0]):.2e} / float32 oracle {np.median(errs['qvel'][1]):.2e}, "
          f"worst {errs['qvel'][0].max():.2e} / {errs['qvel'][1].max():.2e}")
    assert_substep_sample_bounds(errs)


def test_error_paths():
    env, O, cl = make_env_and_oracle(num_envs=8, wrappers=False)
    with pytest.raises(ValueError):
        env.step(None, torch.zeros((3, 3), device=DEV))
    from track_mjx_amd import hip
    import ctypes as C
    h = C.c_void_p()
    assert hip.lib().tmjx_model_create(b"garbage", 7, C.byref(h)) != 0
    assert b"" != hip.lib().tmjx_last_error()


def test_train_entrypoint_smoke():
    """python -m track_mjx_amd.train key=value ... (mirror of track_mjx/train.py:56) runs one PPO training step."""
    from track_mjx_amd import train
    overrides = ["train_setup.train_config.num_envs=256", "train_setup.train_config.batch_size=64",
                 "train_setup.train_config.num_minibatches=4", "train_setup.train_config.unroll_length=5",
                 "train_setup.train_config.num_updates_per_batch=2", "network_config.encoder_layer_sizes=[64,64]",
                 "network_config.decoder_layer_sizes=[64,64]", "network_config.critic_layer_sizes=[64,64]",
                 "train_setup.train_config.num_timesteps=100000", "train_setup.eval_every=50000", "train_setup.reset_every=50000",
                 "max_training_steps=2", "n_synthetic_clips=4"]
    train.main(overrides)


@pytest.mark.gpu
def test_train_calls_policy_params_fn_after_every_eval(tmp_path):
    """ppo.train(..., policy_params_fn=f, checkpoint_callback=c): f is called by process 0 after each eval epoch with the keyword arguments of
    track_mjx/agent/mlp_ppo/ppo.py:762-781 (current_step, jit_logging_inference_fn, params, policy_params_fn_key, render_video by
    env_config.render_interval), the logging policy evaluates the PASSED params deterministically (ppo_networks.py:103-149), and the
    checkpoint callback follows every saved step (ppo.py:713-715,787-795)."""
    import torch
    from track_mjx_amd import config as _config
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.train import build_env
    cfg = _config.default_config()
    cfg["env_config"]["render_interval"] = 2
    dev = torch.device("cuda:0")
    env = build_env(cfg, 64, dev, n_clips=4)
    calls, saved = [], []

    def f(**kw):
        assert set(kw) == {"current_step", "jit_logging_inference_fn", "params", "policy_params_fn_key", "render_video"}
        obs = torch.randn((5, env.observation_size), generator=torch.Generator().manual_seed(kw["current_step"]))
        act, extra = kw["jit_logging_inference_fn"](kw["params"], obs, kw["policy_params_fn_key"])
        act1, extra1 = kw["jit_logging_inference_fn"](kw["params"], obs, kw["policy_params_fn_key"])
        act2, extra2 = kw["jit_logging_inference_fn"](kw["params"], obs, None)
        # ppo_networks.py:116-130: the ACTION is the mode (tanh-squashed), the latent is still sampled from key_network: same key => same action,
        # another key => another latent sample => another action; the latent statistics do not depend on the key
        assert act.shape == (5, 38) and torch.equal(act, act1) and not torch.equal(act, act2) and (act.abs() <= 1).all()
        assert torch.equal(extra["latent_mean"], extra2["latent_mean"]) and torch.equal(extra["latent_logvar"], extra2["latent_logvar"])
        assert extra["latent_mean"].shape == (5, 60) and extra["latent_logvar"].shape == (5, 60)
        calls.append((kw["current_step"], kw["render_video"], kw["policy_params_fn_key"], act.cpu(), {k: v.clone() for k, v in kw["params"][1].items()}))

    def cb(it):          # a failing user callback must not end the run (ppo.py:713-717: try / except + warning)
        saved.append(it)
        if it == 1:
            raise RuntimeError("user callback failed")

    mk, params, metrics = ppo.train(env, num_timesteps=3 * 64 * 4 * 5, episode_length=195, config_dict=cfg, num_evals=4, unroll_length=5, batch_size=64,
                                    num_minibatches=4, num_updates_per_batch=1, normalize_observations=True, encoder_hidden_layer_sizes=(64,),
                                    decoder_hidden_layer_sizes=(64,), value_hidden_layer_sizes=(64,), learning_rate=1e-3, seed=1,
                                    policy_params_fn=f, checkpoint_callback=cb, checkpoint_path=str(tmp_path / "ck"))
    assert [c[0] for c in calls] == [1, 2, 3] and [c[1] for c in calls] == [False, True, False]
    assert len({c[2] for c in calls}) == 3, "a fresh key per call"
    assert saved == [0, 1, 2, 3]
    # the params handed over are those of THAT eval: they change between calls and the last ones are what train() returns
    w1, w3 = calls[0][4], calls[2][4]
    assert any(not torch.equal(w1[k], w3[k]) for k in w1) and all(torch.equal(w3[k], params[1][k]) for k in w3)
    # the same observations through the first call's params give the first call's actions again, whatever the learner holds now
    obs = torch.randn((5, env.observation_size), generator=torch.Generator().manual_seed(1))
    pol = mk(params=None, deterministic=True)
    assert pol(obs.to(dev))[0].shape == (5, 38)


@pytest.mark.gpu
def test_train_resume_continues_env_steps_iteration_and_noise_streams(tmp_path):
    """restore_from= brings back the whole training state (checkpointing.load_training_state, track_mjx/agent/mlp_ppo/ppo.py:561-567): env_steps and
    the iteration continue (new step directories, none overwritten), the device-side Philox counters do not restart at 0."""
    import os
    import numpy as np
    from track_mjx_amd import train
    base = ["train_setup.train_config.num_envs=256", "train_setup.train_config.batch_size=64", "train_setup.train_config.num_minibatches=4",
            "train_setup.train_config.unroll_length=5", "train_setup.train_config.num_updates_per_batch=2", "network_config.encoder_layer_sizes=[64,64]",
            "network_config.decoder_layer_sizes=[64,64]", "network_config.critic_layer_sizes=[64,64]", "train_setup.train_config.num_timesteps=6400",
            "train_setup.eval_every=640", "train_setup.reset_every=640", "n_synthetic_clips=4", "train_setup.train_config.num_eval_envs=0"]
    d = tmp_path / "ck"
    train.main(base + [f"checkpoint_path={d}", "max_training_steps=2"])
    first = sorted(int(x) for x in os.listdir(d))
    assert first == [0, 1, 2], first
    with np.load(d / "2" / "train_state.npz") as z:
        steps2, draws2 = int(z["env_steps"]), int(z["rng/sgd_draw_counter"][0])
        assert int(z["iteration"]) == 2 and steps2 == 2 * 64 * 4 * 5 and draws2 > 0 and int(z["rng/act_counters/0"][0]) > 0
    train.main(base + [f"checkpoint_path={d}", f"restore_from={d}", "max_training_steps=1"])
    assert sorted(int(x) for x in os.listdir(d)) == [0, 1, 2, 3]
    with np.load(d / "3" / "train_state.npz") as z:
        assert int(z["env_steps"]) == steps2 + 64 * 4 * 5 and int(z["iteration"]) == 3 and int(z["rng/sgd_draw_counter"][0]) > draws2
        assert int(z["optimizer_state/count"]) == 3 * 2 * 4


@pytest.mark.gpu
def test_fused_ppo_loss_head_matches_torch_reference():
    """tmjx_ppo_loss (csrc/ppo_kernels.h) against the plain-torch restatement of losses.py:103-245: loss terms and the
    gradient of every network parameter (fp32 both; tolerance 2e-4 of the largest gradient entry)."""
    import torch
    from track_mjx_amd.agent import losses
    from track_mjx_amd.agent.networks import IntentionPolicy, RunningStatistics, ValueNet
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    obs, ref, nu, Z, T, B = 96, 40, 38, 60, 7, 96
    policy = IntentionPolicy(obs, ref, nu, Z, (64, 64), (64, 64)).to(dev)
    value = ValueNet(obs, (64, 64)).to(dev)
    with torch.no_grad():
        policy.head.weight.mul_(0.05)   # keeps the importance ratios O(1) although the latent noise differs between calls
    norm = RunningStatistics(obs, dev)
    g = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
    data = {"observation": rnd(T, B, obs), "next_observation_last": rnd(B, obs), "raw_action": rnd(T, B, nu) * 0.7,
            "log_prob": rnd(T, B) * 0.3 - 20.0, "reward": rnd(T, B).abs(),
            "discount": (torch.rand(T, B, generator=g, device=dev) > 0.1).float(),
            "truncation": (torch.rand(T, B, generator=g, device=dev) > 0.9).float()}
    # behaviour log-probs close to the current policy's so that rho is O(1) and both clip branches occur
    with torch.no_grad():
        from track_mjx_amd.agent.networks import NormalTanh
        lg, _, _ = policy(norm.normalize(data["observation"]))
        data["log_prob"] = NormalTanh.log_prob(lg, data["raw_action"]) + rnd(T, B) * 0.3
    hp = dict(entropy_cost=1e-2, kl_weight=0.1, discounting=0.95, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.2)
    params = list(policy.parameters()) + list(value.parameters())
    torch.manual_seed(99)
    l_ref, m_ref = losses.compute_ppo_loss(policy, value, norm, data, **hp)
    g_ref = torch.autograd.grad(l_ref, params)
    torch.manual_seed(99)
    l_fus, m_fus = losses.compute_ppo_loss_fused(policy, value, norm, data, **hp)
    g_fus = torch.autograd.grad(l_fus, params)
    for k in ("total_loss", "policy_loss", "v_loss", "kl_latent_loss", "entropy_loss"):
        a, b = float(m_fus[k]), float(m_ref[k])
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (k, a, b)
    for a, b in zip(g_fus, g_ref):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-7, (a.shape, float((a - b).abs().max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(7, 96), (20, 1024), (3, 70)])
def test_loss_head_in_phases_gives_the_one_call_forms_gradients(T, B):
    """tmjx_ppo_loss_phases — A on the policy's stream, B on the value network's, C after both, D beside the backward pass — against tmjx_ppo_loss on the same
    inputs: the three gradient arrays bit for bit, the eight scalars to 1e-6 (the entropy / KL sums are added up in D instead of B); a second pass with
    cfg.accumulate adds to `out`; an unroll longer than 24 steps is refused."""
    import ctypes as C
    import torch
    from track_mjx_amd import hip
    dev = torch.device("cuda:0")
    L = hip.lib()
    g = torch.Generator(device=dev).manual_seed(T * 1000 + B)
    A, Z = 38, 60
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
    logits, raw, noise, fc2 = rnd(T, B, 2 * A) * 0.5, rnd(T, B, A) * 0.7, rnd(T, B, A), rnd(T, B, 2 * Z) * 0.3
    blogp, baseline, bootstrap, reward = rnd(T, B) * 0.3 - 30.0, rnd(T, B), rnd(B), rnd(T, B).abs()
    discount = (torch.rand(T, B, generator=g, device=dev) > 0.1).float()
    trunc = (torch.rand(T, B, generator=g, device=dev) > 0.9).float()
    ins = [logits, raw, blogp, noise, baseline, bootstrap, reward, discount, trunc, fc2]
    cfg = hip.PpoCfg(T, B, A, Z, 1.0, 0.95, 0.95, 0.2, 1e-2, 0.1, 1, 0)

    def run(phased: bool, accumulate: int = 0, out=None):
        cfg.accumulate = accumulate
        outs = [torch.empty_like(logits), torch.empty_like(baseline), torch.empty_like(fc2), torch.zeros(L.tmjx_ppo_scratch_floats(T, B), device=dev),
                torch.zeros(8, device=dev) if out is None else out]
        ptr = [C.c_void_p(a.data_ptr()) for a in ins + outs]
        cur, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
        if not phased:
            hip.check(L.tmjx_ppo_loss(C.byref(cfg), *ptr, C.c_void_p(cur.cuda_stream)), "tmjx_ppo_loss")
        else:
            side.wait_stream(cur)
            hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *ptr, 2, C.c_void_p(side.cuda_stream)), "B")
            hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *ptr, 1, C.c_void_p(cur.cuda_stream)), "A")
            cur.wait_stream(side)
            hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *ptr, 4, C.c_void_p(cur.cuda_stream)), "C")
            side.wait_stream(cur)
            hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *ptr, 8, C.c_void_p(side.cuda_stream)), "D")
            cur.wait_stream(side)
        torch.cuda.synchronize()
        return outs
    ref, got = run(False), run(True)
    for k in range(3):
        assert torch.equal(ref[k], got[k]), k
    # A and C in one call = ONE launch (k_ppo_ac: the two bodies back to back; what the learner issues behind B): the same bits again, scalars included
    cfg.accumulate = 0
    ac = [torch.empty_like(logits), torch.empty_like(baseline), torch.empty_like(fc2), torch.zeros(L.tmjx_ppo_scratch_floats(T, B), device=dev), torch.zeros(8, device=dev)]
    pac = [C.c_void_p(a.data_ptr()) for a in ins + ac]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for mask in (2, 1 | 4, 8):
        hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *pac, mask, st), "B | AC | D")
    torch.cuda.synchronize()
    for k in (0, 1, 2, 4):
        assert torch.equal(ac[k], got[k]), k
    assert float((ref[4] - got[4]).abs().max()) <= 1e-6 * max(1.0, float(ref[4].abs().max())), (ref[4], got[4])
    one = run(True, 0)[4]
    twice = run(True, 1, out=one.clone())[4]
    assert float((twice - 2 * one).abs().max()) <= 1e-6 * max(1.0, float(one.abs().max()))
    all_in_one = [torch.empty_like(logits), torch.empty_like(baseline), torch.empty_like(fc2), torch.zeros(L.tmjx_ppo_scratch_floats(T, B), device=dev), torch.zeros(8, device=dev)]
    cfg.accumulate = 0
    hip.check(L.tmjx_ppo_loss_phases(C.byref(cfg), *[C.c_void_p(a.data_ptr()) for a in ins + all_in_one], 15, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ABCD")
    torch.cuda.synchronize()
    assert torch.equal(all_in_one[0], ref[0]) and torch.equal(all_in_one[4], got[4])
    long = hip.PpoCfg(25, B, A, Z, 1.0, 0.95, 0.95, 0.2, 1e-2, 0.1, 1, 0)
    assert L.tmjx_ppo_loss_phases(C.byref(long), *[C.c_void_p(a.data_ptr()) for a in ins + all_in_one], 15, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("H", [64, 256, 1024])
def test_fused_silu_layernorm_block_matches_torch(H):
    """tmjx_silu_ln_fwd / _bwd against torch's linear -> silu -> layer_norm, outputs and all four gradients."""
    import torch
    import torch.nn.functional as F
    from track_mjx_amd.agent.networks import _SiluLayerNormFn
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(H)
    rows = 1000
    z = torch.randn(rows, H, generator=g, device=dev, requires_grad=True)
    bias = (torch.randn(H, generator=g, device=dev) * 0.3).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(H, generator=g, device=dev)).requires_grad_()
    beta = (0.1 * torch.randn(H, generator=g, device=dev)).requires_grad_()
    up = torch.randn(rows, H, generator=g, device=dev)
    y_ref = F.layer_norm(F.silu(z + bias), (H,), gamma, beta, 1e-6)
    g_ref = torch.autograd.grad((y_ref * up).sum(), (z, bias, gamma, beta))
    y = _SiluLayerNormFn.apply(z, bias, gamma, beta, 1e-6)
    g_fus = torch.autograd.grad((y * up).sum(), (z, bias, gamma, beta))
    assert float((y - y_ref).abs().max()) < 2e-5
    for a, b in zip(g_fus, g_ref):
        assert float((a - b).abs().max()) <= 1e-4 * (float(b.abs().max()) + 1e-6), (a.shape, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.gpu
def test_gather_normalize_matches_torch():
    import torch
    from track_mjx_amd.agent import losses
    from track_mjx_amd.agent.networks import RunningStatistics
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2)
    src = torch.randn(5, 300, 696, generator=g, device=dev)
    idx = torch.randint(0, 300, (77,), generator=g, device=dev)
    norm = RunningStatistics(696, dev)
    norm.update(src)
    got = losses.gather_normalize(src, idx, norm)
    ref = norm.normalize(src.index_select(1, idx))
    assert torch.equal(got, ref)
    assert torch.equal(losses.gather_normalize(src[0], idx, norm), norm.normalize(src[0].index_select(0, idx)))


@pytest.mark.gpu
def test_fused_minibatch_gather_matches_index_select():
    """tmjx_gather_minibatch (every leaf of the roll-out data at the minibatch rows, one launch) is bit-identical to index_select per leaf
    + the normaliser (ppo.py:304-317: x[perm] reshaped to minibatches, the same rows for every leaf)."""
    import torch
    from track_mjx_amd.agent import losses
    from track_mjx_amd.agent.networks import RunningStatistics
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    T, R, W, A, B = 5, 300, 696, 38, 77
    buf = {"observation": torch.randn(T, R, W, generator=g, device=dev), "raw_action": torch.randn(T, R, A, generator=g, device=dev),
           "log_prob": torch.randn(T, R, generator=g, device=dev), "reward": torch.randn(T, R, generator=g, device=dev),
           "discount": torch.rand(T, R, generator=g, device=dev), "truncation": torch.rand(T, R, generator=g, device=dev),
           "next_observation_last": torch.randn(R, W, generator=g, device=dev)}
    idx = torch.randint(0, R, (B,), generator=g, device=dev)
    norm = RunningStatistics(W, dev)
    norm.update(buf["observation"])
    got = losses.gather_minibatch(buf, idx, norm)
    assert torch.equal(got["observation_normalized"], norm.normalize(buf["observation"].index_select(1, idx)))
    assert torch.equal(got["next_observation_last_normalized"], norm.normalize(buf["next_observation_last"].index_select(0, idx)))
    for k in ("raw_action", "log_prob", "reward", "discount", "truncation"):
        assert torch.equal(got[k], buf[k].index_select(1, idx)), k


@pytest.mark.gpu
def test_evaluator_first_episode_sums():
    """agent/evaluator.py (brax acting.Evaluator + EvalWrapper semantics, ppo.py:83-124,629-668): episode_* are sums over the FIRST
    episode of each env only, avg_episode_length counts its steps — checked against a numpy accumulation of the same roll-out."""
    from track_mjx_amd.agent.evaluator import EvalWrapper, Evaluator
    n = 32
    env, _, _ = make_env_and_oracle(num_envs=n, n_clips=4, wrappers=True, episode_length=25)
    g = torch.Generator(device="cuda").manual_seed(4)
    acts = [(torch.randn((38, n), generator=g, device="cuda") * 0.6).clamp(-1, 1) for _ in range(40)]
    w = EvalWrapper(env)
    st = w.reset(torch.Generator().manual_seed(0))
    active = np.ones(n); ep_rew = np.zeros(n); ep_len = np.zeros(n); ep_pos = np.zeros(n)
    for a in acts:
        st = w.step(st, a)
        rew, done = st.reward.cpu().numpy().astype(np.float64), st.done.cpu().numpy()
        steps = st.info["steps"].cpu().numpy()
        ep_len = np.where(active > 0, steps, ep_len)
        ep_rew += rew * active; ep_pos += st.metrics["pos_reward"].cpu().numpy() * active
        active = active * (1 - done)
    assert (active == 0).all(), "every env must finish its first episode (episode_length 25 < 40 steps)"
    assert np.allclose(w.episode_metrics["reward"].cpu().numpy(), ep_rew, rtol=1e-5, atol=1e-6, equal_nan=True)
    # (an env whose physics blew up carries NaN metrics through the episode sums, exactly as brax's EvalWrapper does)
    assert np.allclose(w.episode_metrics["pos_reward"].cpu().numpy(), ep_pos, rtol=1e-5, atol=1e-6, equal_nan=True)
    assert np.array_equal(w.episode_steps.cpu().numpy(), ep_len) and ep_len.max() <= 25 and ep_len.min() >= 1
    ev = Evaluator(env, lambda obs: (acts[0], None), episode_length=25, seed=1)
    m = ev.run_evaluation({"training/sps": 1.0})
    assert m["training/sps"] == 1.0 and "eval/episode_reward" in m and "eval/episode_reward_std" in m
    assert 1 <= m["eval/avg_episode_length"] <= 25 and m["eval/sps"] > 0


@pytest.mark.gpu
def test_many_active_rows_multi_slot_gpu():
    """All 67 joint limits violated and the body sunk into the floor: > 64 active constraint rows, i.e. the multi-slot path of the
    compacted rows (ballot / mbcnt prefix counts on the device) against the float64 oracle."""
    n = 8
    env, _, cl = make_env_and_oracle(num_envs=n, wrappers=False)
    O64 = make_oracle(env._blob, cl, "f64")
    rng = np.random.default_rng(5)
    qpos, qvel = _init_states(cl, n, rng)
    rngs = np.asarray(env.walker.model["jnt_range"]).reshape(-1, 2)[1:]
    for e in range(n):
        over = rng.uniform(0.005, 0.03, size=67) * (rngs[:, 1] - rngs[:, 0])
        qpos[e, 7:] = np.where(rng.random(67) < 0.5, rngs[:, 1] + over, rngs[:, 0] - over)
        qpos[e, 2] -= 0.02 * (e + 1)
    act = rng.uniform(-0.1, 0.1, size=(n, 38))
    env.rows("qpos").copy_(torch.from_numpy(qpos.T.astype(np.float32)))
    env.rows("qvel").copy_(torch.from_numpy(qvel.T.astype(np.float32)))
    env.rows("act").copy_(torch.from_numpy(act.T.astype(np.float32)))
    env.forward()
    torch.cuda.synchronize()
    ds = []
    for e in range(n):
        d = O64.new_data(qpos[e], qvel[e]); O64.set(d, "act", act[e]); O64.forward(d); ds.append(d)
    for name, tol in (("efc_D", 5e-5), ("efc_aref", 5e-5), ("qacc_smooth", 5e-4), ("efc_force", 5e-3), ("qacc", 5e-3)):
        got = env.rows(name).cpu().numpy().astype(np.float64)
        ref = np.stack([O64.get(d, name) for d in ds], 1)
        assert rel_err(got, ref) < tol, (name, rel_err(got, ref))


@pytest.mark.gpu
def test_fused_inference_tail_matches_torch_path():
    """PPOLearner._act_fused (tmjx_latent_concat + tmjx_sample_action) against the plain torch inference path with the same
    generator state: same action, raw action, log-prob."""
    from track_mjx_amd.agent import ppo
    env, _, _ = make_env_and_oracle(num_envs=64, n_clips=4, wrappers=True)
    L = ppo.PPOLearner(env, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5,
                       batch_size=16, num_minibatches=4, num_updates_per_batch=1, seed=3, use_graph=False, act_rng="torch")
    st = env.reset(torch.Generator().manual_seed(0))
    state = L.gen.get_state()
    a_f, e_f = L.act(st.obs)
    L.gen.set_state(state)
    x = L.normalizer.normalize(st.obs)
    eps = torch.randn((64, 60), generator=L.gen, device=L.dev)
    logits, mean, logvar = L.policy(x, eps=eps)
    noise = torch.randn((64, 38), generator=L.gen, device=L.dev)
    from track_mjx_amd.agent.networks import NormalTanh
    raw = NormalTanh.sample_no_postprocessing(logits, noise)
    # (the kernels' SiLU is v_exp_f32 + v_rcp_f32, csrc/silu_math.h: ~1e-6 relative per activation against torch's expf + division, a few 1e-6 after four blocks)
    assert torch.allclose(e_f["raw_action"], raw, rtol=1e-4, atol=1e-5), float((e_f["raw_action"] - raw).abs().max())
    assert torch.allclose(a_f, torch.tanh(raw), rtol=1e-4, atol=1e-5)
    assert torch.allclose(e_f["log_prob"], NormalTanh.log_prob(logits, raw), rtol=1e-4, atol=1e-3)
    assert torch.allclose(e_f["latent_mean"], mean, rtol=1e-4, atol=1e-5)
    # the LDS-free variant (tmjx_linear_nolds for every dense layer, normaliser folded into the first layer / the concat)
    L.normalizer.update(st.obs.reshape(1, 64, -1) * 1.0)
    x = L.normalizer.normalize(st.obs)
    L.gen.set_state(state)
    eps = torch.randn((64, 60), generator=L.gen, device=L.dev)
    logits, mean, logvar = L.policy(x, eps=eps)
    noise = torch.randn((64, 38), generator=L.gen, device=L.dev)
    raw = NormalTanh.sample_no_postprocessing(logits, noise)
    L.lds_free = True
    L.gen.set_state(state)
    a_l, e_l = L.act(st.obs)
    assert torch.allclose(e_l["logits"], logits, rtol=1e-3, atol=2e-4), float((e_l["logits"] - logits).abs().max())
    assert torch.allclose(e_l["raw_action"], raw, rtol=1e-3, atol=3e-4)
    assert torch.allclose(a_l, torch.tanh(raw), rtol=1e-3, atol=3e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [(64, 64), (48, 44, 36)])
def test_pipelined_rollout_env_groups(sizes):
    """PPOLearner with a LIST of envs (bench.py --pipeline N; the default is three groups, of unequal size when the env count does not divide —
    ppo.group_sizes): the groups' roll-outs run on separate HIP streams with the LDS-free inference.  The roll-out buffer must hold each
    group's rows in its own slice, and a full training step must stay finite."""
    from track_mjx_amd.agent import ppo
    envs = [make_env_and_oracle(num_envs=n, n_clips=4, wrappers=True, seed=k)[0] for k, n in enumerate(sizes)]
    L = ppo.PPOLearner(envs, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5,
                       batch_size=32, num_minibatches=4, num_updates_per_batch=2, seed=3)
    assert L.lds_free and L.n_local == 128 and L.unrolls == 1
    first = []
    for k, e in enumerate(envs):
        L.states[k] = e.reset(torch.Generator().manual_seed(10 + k))
        first.append(L.states[k].obs.clone())
    before = L.grads.params[0].detach().clone()
    L.collect()
    torch.cuda.synchronize()
    offs = [sum(sizes[:k]) for k in range(len(sizes) + 1)]
    for k in range(len(sizes)):
        assert torch.equal(L.buf["observation"][0, offs[k]:offs[k + 1]], first[k])
        assert torch.equal(L.buf["next_observation_last"][offs[k]:offs[k + 1]], L.states[k].obs)
    for name, v in L.buf.items():
        assert torch.isfinite(v).all(), name
    m = L.update(1)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in m.values())
    assert not torch.equal(before, L.grads.params[0].detach())


@pytest.mark.gpu
def test_rollout_store_kernel_and_multi_unroll_buffer_rows():
    """tmjx_rollout_store (one launch per env-group step instead of seven copies) against the torch ops it replaces, and — through a
    learner with TWO unrolls per training step — that row 0 of the second unroll equals next_observation_last of the first, every
    observation row t + 1 is the env's observation after step t, and the env groups share ONE resident clip table."""
    import ctypes as C
    from track_mjx_amd import hip
    from track_mjx_amd.agent import ppo
    g = torch.Generator(device=DEV).manual_seed(0)
    n, W, A = 200, 696, 38
    obs = torch.randn((W, n), generator=g, device=DEV)
    raw, logp = torch.randn((n, A), generator=g, device=DEV), torch.randn(n, generator=g, device=DEV)
    rew, done, tr = torch.randn(n, generator=g, device=DEV), (torch.rand(n, generator=g, device=DEV) < 0.3).float(), (torch.rand(n, generator=g, device=DEV) < 0.1).float()
    d0, d1 = torch.zeros((n, W), device=DEV), torch.zeros((n, W), device=DEV)
    o = [torch.zeros((n, A), device=DEV)] + [torch.zeros(n, device=DEV) for _ in range(4)]
    p = lambda t: t.data_ptr()  # noqa: E731
    q = hip.RolloutStore(p(obs), p(d0), p(d1), p(raw), p(o[0]), p(logp), p(o[1]), p(rew), p(o[2]), p(done), p(o[3]), p(tr), p(o[4]), n, W, A)
    hip.check(hip.lib().tmjx_rollout_store(C.byref(q), None), "tmjx_rollout_store")
    torch.cuda.synchronize()
    assert torch.equal(d0, obs.t()) and torch.equal(d1, obs.t()) and torch.equal(o[0], raw) and torch.equal(o[1], logp)
    assert torch.equal(o[2], rew) and torch.equal(o[3], 1 - done) and torch.equal(o[4], tr)
    q = hip.RolloutStore(None, None, None, None, None, None, None, p(rew), p(o[1]), None, None, None, None, n, 0, 0)      # every destination is optional
    hip.check(hip.lib().tmjx_rollout_store(C.byref(q), None), "tmjx_rollout_store")
    torch.cuda.synchronize()
    assert torch.equal(o[1], rew)
    # the third observation destination alone (the acting policy's row-major staging copy of the new observation)
    d2 = torch.zeros((n, W), device=DEV)
    q = hip.RolloutStore(p(obs), None, None, None, None, None, None, None, None, None, None, None, None, n, W, 0, p(d2))
    hip.check(hip.lib().tmjx_rollout_store(C.byref(q), None), "tmjx_rollout_store")
    torch.cuda.synchronize()
    assert torch.equal(d2, obs.t())
    # two env groups sharing one clip table, two unrolls per training step
    e0 = make_env_and_oracle(num_envs=32, n_clips=4, wrappers=True, seed=0)[0]
    from track_mjx_amd.environment import MultiClipTracking, RewardConfig, wrap
    from tests.common import default_walker
    w, cfg = default_walker()
    e1 = wrap(MultiClipTracking(e0._reference_clips, w, RewardConfig(**cfg["env_config"]["reward_weights"]), **cfg["env_config"]["env_args"],
                                **cfg["reference_config"], num_envs=32, device=DEV, share_clips_with=e0), episode_length=195)
    L = ppo.PPOLearner([e0, e1], encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=4,
                       batch_size=32, num_minibatches=4, num_updates_per_batch=1, seed=3)
    assert L.unrolls == 2 and L.n_local == 64
    seen = []
    orig = L._store_transition

    def spy(env, st, extra, d0_, d1_, t, sl, d2_=None):
        seen.append((t, sl.start, st.obs.clone(), st.reward.clone(), st.done.clone(), extra["raw_action"].clone()))
        orig(env, st, extra, d0_, d1_, t, sl, d2_)
        assert d2_ is not None          # the acting policy's row-major staging copy of the group's new observation
    L._store_transition = spy
    for k, e in enumerate([e0, e1]):
        L.states[k] = e.reset(torch.Generator().manual_seed(20 + k), torch.arange(32, dtype=torch.int32) % 4)
    L.collect()
    torch.cuda.synchronize()
    b = L.buf
    assert torch.equal(b["observation"][0, 64:128], b["next_observation_last"][0:64])      # unroll 1 starts where unroll 0 ended
    for t, start, ob, rw, dn, ra in seen:
        sl = slice(start, start + 32)
        assert torch.equal(b["reward"][t, sl], rw) and torch.equal(b["discount"][t, sl], 1 - dn) and torch.equal(b["raw_action"][t, sl], ra)
        nxt = b["observation"][t + 1, sl] if t + 1 < 4 else b["next_observation_last"][sl]
        assert torch.equal(nxt, ob)
    assert len(seen) == 2 * 4 * 2
    # the staging copies hold every group's newest observation, row-major: what the next inference reads (LDS-free acting path)
    for k in range(2):
        assert L._obs_rm is not None and torch.equal(L._obs_rm[k], L.states[k].obs)
    out = L.update(0)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values())


@pytest.mark.gpu
def test_bf16_gemm_inputs_match_fp32_gradients():
    """BASELINE config 5: MLP GEMMs with bf16 inputs / fp32 accumulation (autocast around the networks, parameters, loss head,
    optimizer in fp32).  On the same roll-out buffer and parameters the loss and the flat gradient must agree with the fp32
    learner to bf16 resolution (loss 2 %, gradient cosine > 0.98), through the eager path and through the captured hipGraph."""
    from track_mjx_amd.agent import ppo
    env = make_env_and_oracle(num_envs=256, n_clips=4, wrappers=True)[0]
    kw = dict(encoder_layers=(256, 256), decoder_layers=(256, 256), critic_layers=(256, 256), latents=60, unroll_length=20,
              batch_size=256, num_minibatches=4, num_updates_per_batch=1, seed=5)
    L32 = ppo.PPOLearner(env, use_graph=False, **kw)
    L16 = ppo.PPOLearner(env, use_graph=False, matmul_dtype=torch.bfloat16, **kw)
    L32.states[0] = env.reset(torch.Generator().manual_seed(1))
    L32.collect()
    for k in L32.buf:
        L16.buf[k].copy_(L32.buf[k])
    for n in (L32, L16):
        n.normalizer.update(n.buf["observation"])
    idx = torch.arange(L32.local_batch, device=env.device)
    torch.manual_seed(0)
    m32 = L32._minibatch_grads(idx, 0.1)
    g32 = L32.grads.flat.clone()
    torch.manual_seed(0)
    m16 = L16._minibatch_grads(idx, 0.1)
    g16 = L16.grads.flat.clone()
    assert torch.isfinite(g16).all() and g16.dtype == torch.float32
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    print(f"\nbf16 GEMM-input mode against fp32: gradient cosine {cos:.5f}, loss {float(m16[0]):.6f} vs {float(m32[0]):.6f}")
    assert cos > 0.98, cos
    assert abs(float(m16[0]) - float(m32[0])) <= 0.02 * abs(float(m32[0])) + 1e-3, (m16, m32)
    # the captured-graph path of a full update (what bench.py --config cfg5 runs)
    L16.use_graph = True
    out = L16.update(0)
    torch.cuda.synchronize()
    assert L16._graph is not None, "hipGraph capture of the bf16 SGD step failed"
    assert all(bool(torch.isfinite(v).all()) for v in out.values())


@pytest.mark.gpu
def test_full_size_networks_training_step():
    """BASELINE config 4 (rodent-mc-intention nets of rodent-full-clips.yaml:50-57): parameter count, one training step finite,
    fused block path (widths 1024 / 512 / 256) equal to the torch composition."""
    from track_mjx_amd.agent import ppo
    env = make_env_and_oracle(num_envs=128, n_clips=4, wrappers=True)[0]
    L = ppo.PPOLearner(env, encoder_layers=(1024, 512, 512, 512, 512), decoder_layers=(512, 512, 512, 256, 256),
                       critic_layers=(512, 512, 512, 512, 512, 256), latents=60, unroll_length=20, batch_size=128, num_minibatches=2,
                       num_updates_per_batch=1, seed=2)
    assert L.n_params() == 4_294_853, L.n_params()     # 4.29 M parameters
    L.states[0] = env.reset(torch.Generator().manual_seed(3))
    out = L.training_step(0)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values())
    x = torch.randn(512, 470, device=env.device)
    with torch.no_grad():
        h_fused = L.policy.encoder(x)
        h_ref = x
        for blk in L.policy.encoder:
            h_ref = blk.norm(torch.nn.functional.silu(torch.nn.functional.linear(h_ref, blk.dense.weight, blk.dense.bias)))
    assert float((h_fused - h_ref).abs().max()) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_baseline_configs_4_and_5_at_full_size(cfg):
    """BASELINE configs[3] / configs[4] as bench.py builds them, at FULL size, one training step each through the product path: config 4 =
    rodent-mc-intention nets (rodent-full-clips.yaml:50-57) at 4096 envs in two pipelined groups, fp32; config 5 = 8192 envs, a 1024-clip table
    (per-env clip gather), the same nets with bf16 GEMM inputs on the repo's own kernels.  Finite losses, moved parameters, and for config 5 no
    library GEMM: the bf16 shadows exist and every hidden activation of the chains is bf16."""
    import bench
    from track_mjx_amd import clips as _clips, config as _config
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.environment import wrap
    from track_mjx_amd.train import build_env
    from track_mjx_amd.walker import Rodent
    bc = bench.CONFIGS[cfg]
    c = _config.default_config()
    c["network_config"].update(**bc["nets"])
    tc, nc = c["train_setup"]["train_config"], c["network_config"]
    n = bc["envs_per_gpu"]
    table = _clips.make_synthetic_clips(Rodent(**c["walker_config"]).model, bc["n_clips"], n_frames=c["reference_config"]["clip_length"], mocap_hz=c["env_config"]["env_args"]["mocap_hz"])
    sizes = ppo.group_sizes(n, 3)               # three env groups, as bench.py builds them
    e0 = wrap(build_env(c, sizes[0], DEV, reference_clip=table), episode_length=195)
    envs = [e0] + [wrap(build_env(c, sz, DEV, reference_clip=table, share_clips_with=e0), episode_length=195) for sz in sizes[1:]]
    L = ppo.PPOLearner(envs, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                       latents=nc["intention_size"], unroll_length=tc["unroll_length"], batch_size=tc["batch_size"] * n // 4096, num_minibatches=tc["num_minibatches"],
                       num_updates_per_batch=1, kl_weight=nc["kl_weight"], seed=0, matmul_dtype=torch.bfloat16 if bc["matmul_dtype"] == "bf16" else None)
    assert L.n_params() == 4_294_853 and L.local_batch * L.T == (20480 if cfg == "cfg4" else 40960)
    g = torch.Generator().manual_seed(1)
    for k, e in enumerate(envs):
        idx = torch.randint(0, bc["n_clips"], (sizes[k],), generator=g, dtype=torch.int32)
        L.states[k] = e.reset(g, idx)
        assert int(e.istate_buf[e.layout.i_clip_idx].max()) < bc["n_clips"]
    before = L.opt.flat.clone()
    out = L.training_step(0)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values()), out
    assert float((L.opt.flat - before).abs().max()) > 0
    if cfg == "cfg5":
        assert L.shadows is not None and all(w.dtype == torch.bfloat16 for w in L.shadows.w.values())
        assert int(envs[0].istate_buf[envs[0].layout.i_clip_idx].max()) > 64          # the 1024-clip table is really indexed
    # the physics at this launch size against the oracle: after the training step's 20 control steps, single-substep launches of all envs of the
    # three groups, a sample of envs (both ends of every launch + a spread) teacher-forced through the float64 / float32 oracles
    from tests.common import assert_substep_sample_bounds, oracle_substep_sample
    O32, O64 = make_oracle(e0._blob, None, "f32"), make_oracle(e0._blob, None, "f64")      # (physics only: no clip table)
    errs = oracle_substep_sample(envs, O32, O64, np.random.default_rng(7), scale=0.3, per_group=22, substeps=2)
    print(f"\n{cfg}: {n} envs in groups {sizes}: teacher-forced substeps on {len(errs['qvel'][0])} env-substeps: qvel median HIP {np.median(errs['qvel'][0]):.2e} / "
          f"float32 oracle {np.median(errs['qvel'][1]):.2e}")
    assert_substep_sample_bounds(errs)


@pytest.mark.gpu
def test_reset_from_jax_key_draws():
    """reset(key) with a jax PRNG key: clip index, start frame and noise are the threefry draws of the reference's reset
    (bit-exact integers; the state equals the reference frame + that noise)."""
    from track_mjx_amd import jax_random as jr
    env, O, cl = make_env_and_oracle(num_envs=64, n_clips=4, wrappers=True)
    key = jr.PRNGKey(42)
    st = env.reset(key)
    torch.cuda.synchronize()
    keys = jr.split(key, 64)
    ci, sf, qn, vn = jr.reset_draws_batch(keys, 4, 74, 73, env._reset_noise_scale)
    assert np.array_equal(st.info["clip_idx"].cpu().numpy(), ci) and np.array_equal(st.info["start_frame"].cpu().numpy(), sf)
    L = env.layout
    qpos = env.state_buf[L.qpos:L.qpos + L.nq].cpu().numpy()
    ref = np.concatenate([cl.position[ci, sf], cl.quaternion[ci, sf], cl.joints[ci, sf]], axis=-1).T
    want = ref + qn
    want[3:7] /= np.linalg.norm(want[3:7], axis=0, keepdims=True)      # the forward pass stores the normalised root quaternion (MJX kinematics)
    assert np.abs(qpos - want).max() < 1e-6
    qvel = env.state_buf[L.qvel:L.qvel + L.nv].cpu().numpy()
    assert np.array_equal(qvel, vn) and np.array_equal(vn, qn[:73])      # the reference draws both from rng1
    st2 = env.reset(keys)           # per-env keys give the same state
    torch.cuda.synchronize()
    assert np.array_equal(env.state_buf[L.qvel:L.qvel + L.nv].cpu().numpy(), vn)


@pytest.mark.gpu
def test_config5_clip_table_gather_bit_exact():
    """BASELINE config 5: 8192 envs on a 1024-clip table (206 MB of body positions alone: a real HBM gather).  With zero reset
    noise the reset state must be the gathered clip frame bit for bit (positions, joints; the root quaternion normalised), the
    reference half of the observation must be finite, and one control step must leave the integer frame bookkeeping exact."""
    n, C = 8192, 1024
    env, O, cl = make_env_and_oracle(num_envs=n, n_clips=C, wrappers=True)
    rng = np.random.default_rng(2)
    ci = rng.integers(0, C, n).astype(np.int32)
    sf = rng.integers(0, 44, n).astype(np.int32)
    z74, z73 = torch.zeros((74, n)), torch.zeros((73, n))
    st = env.reset(None, torch.from_numpy(ci), start_frame=torch.from_numpy(sf), qpos_noise=z74, qvel_noise=z73)
    torch.cuda.synchronize()
    L = env.layout
    qpos = env.state_buf[L.qpos:L.qpos + L.nq].cpu().numpy()
    assert np.array_equal(qpos[:3], cl.position[ci, sf].T) and np.array_equal(qpos[7:], cl.joints[ci, sf].T)
    q = cl.quaternion[ci, sf].T
    assert np.abs(qpos[3:7] - q / np.linalg.norm(q, axis=0, keepdims=True)).max() < 1e-6
    assert np.array_equal(st.info["clip_idx"].cpu().numpy(), ci) and np.array_equal(st.info["start_frame"].cpu().numpy(), sf)
    assert torch.isfinite(st.obs).all()
    # at reset the tracked root position offset of the first trajectory frame is the clip's own displacement between two frames,
    # rotated by the root quaternion: zero noise => |obs[:3]| is bounded by that displacement (0.1 m over 250 frames + z jitter)
    assert float(st.obs[:, :15].abs().max()) < 0.05
    st = env.step(st, torch.zeros((38, n), device=DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(st.obs).all() and torch.isfinite(st.reward).all()
    assert np.array_equal(st.info["clip_idx"].cpu().numpy(), ci)


@pytest.mark.gpu
def test_fused_clip_adam_kernel_matches_torch():
    """tmjx_adam_clip (optax clip_by_global_norm -> adam on the flat buffers) against torch.optim.Adam + clip_grad_norm_."""
    from tests.test_learner_math import _flat_adam_vs_torch
    _flat_adam_vs_torch("cuda:0")


@pytest.mark.gpu
def test_train_from_hdf5_clip_file(tmp_path):
    """SURVEY §8 f2 end to end: synthetic clips exported in the stac-mjx HDF5 layout, loaded through io/load.py by the train
    entrypoint's clip selection (random train/test split), envs built on the training clips, one PPO training step finite; the
    env on the file's clips steps exactly like the env on the in-memory clips."""
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import make_clips_h5
    from track_mjx_amd import clips as _clips, config as _config
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.environment import wrap
    from track_mjx_amd.train import build_env, load_clip_sets
    from track_mjx_amd.walker import Rodent
    cfg = _config.default_config()
    cl = _clips.make_synthetic_clips(Rodent(**cfg["walker_config"]).model, 10)
    make_clips_h5.export(cl, tmp_path / "clips.h5")
    cfg["data_path"] = str(tmp_path / "clips.h5")
    cfg["train_setup"]["train_subset_ratio"] = None
    allc, none = load_clip_sets(cfg)
    assert none is None and np.array_equal(allc.joints, cl.joints)
    e_file, e_mem = build_env(cfg, 32, DEV, reference_clip=allc), build_env(cfg, 32, DEV, reference_clip=cl)
    g1, g2 = torch.Generator().manual_seed(4), torch.Generator().manual_seed(4)
    s1, s2 = e_file.reset(g1), e_mem.reset(g2)
    a = (torch.randn((38, 32), generator=g1) * 0.1).clamp(-1, 1).to(DEV)
    s1, s2 = e_file.step(s1, a), e_mem.step(s2, a)
    torch.cuda.synchronize()
    assert torch.equal(s1.obs, s2.obs) and torch.equal(s1.reward, s2.reward)
    cfg["train_setup"]["train_subset_ratio"] = 0.8
    np.random.seed(0)
    train, test = load_clip_sets(cfg)
    assert train.position.shape[0] == 9 and test.position.shape[0] == 1      # int(10 * (1 - 0.8)) = 1 in floating point, as in the reference
    env = wrap(build_env(cfg, 128, DEV, reference_clip=train), episode_length=195)
    L = ppo.PPOLearner(env, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5, batch_size=128,
                       num_minibatches=2, num_updates_per_batch=1, seed=1)
    L.states[0] = env.reset(torch.Generator().manual_seed(2))
    out = L.training_step(0)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(v).all()) for v in out.values())


@pytest.mark.gpu
def test_loss_head_outside_autograd_gives_the_same_gradients():
    """PPOLearner's path (losses.ppo_loss_and_output_grads: tmjx_ppo_loss outside autograd + one autograd.grad from the network
    outputs with grad_outputs) against the autograd.Function path (compute_ppo_loss_fused + autograd.grad(loss))."""
    from track_mjx_amd.agent import losses as LS, ppo
    env = make_env_and_oracle(num_envs=128, n_clips=4, wrappers=True)[0]
    L = ppo.PPOLearner(env, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=20, batch_size=128,
                       num_minibatches=2, num_updates_per_batch=1, seed=7, use_graph=False)
    L.states[0] = env.reset(torch.Generator().manual_seed(3))
    L.collect()
    L.normalizer.update(L.buf["observation"])
    idx = torch.arange(L.local_batch, device=env.device)
    data = {k: (v.index_select(1, idx) if k != "next_observation_last" else v.index_select(0, idx)) for k, v in L.buf.items()}
    torch.manual_seed(0)
    loss, m_ref = LS.compute_ppo_loss_fused(L.policy, L.value, L.normalizer, data, kl_weight=0.1, **L.hp)
    g_ref = torch.autograd.grad(loss, L.grads.params)
    torch.manual_seed(0)
    m, outs, gouts, out8 = LS.ppo_loss_and_output_grads(L.policy, L.value, L.normalizer, data, kl_weight=0.1, **L.hp)
    g = torch.autograd.grad(outs, L.grads.params, grad_outputs=gouts)
    for a, b in zip(g, g_ref):
        assert torch.equal(a, b)
    for k in m:
        assert float(m[k]) == float(m_ref[k])
    torch.manual_seed(0)
    vec = L._minibatch_grads(idx, 0.1)
    # (the learner runs the loss head in PHASES, whose last launch adds phase A's per-block entropy / KL partials up in another order than the one-call form's
    # records: the gradients are bit-identical — test_loss_head_in_phases_gives_the_one_call_forms_gradients —, the eight scalars agree to 1e-6 as asserted there)
    for x, k in zip(vec, L.METRIC_KEYS):
        assert abs(float(x) - float(m_ref[k])) <= 1e-6 * max(1.0, abs(float(m_ref[k]))), (k, float(x), float(m_ref[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(20480, 76), (20480, 256), (40960, 120), (1025, 1), (4099, 33)])
def test_colsum_kernel_matches_torch(shape):
    """tmjx_colsum (bias gradients of the dense layers) against a float64 column sum."""
    from track_mjx_amd.agent.networks import _colsum
    x = torch.randn(shape, device=DEV)
    got = _colsum(x)
    ref = x.double().sum(0)
    assert got.shape == (shape[1],) and float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max() + shape[0] ** 0.5)


@pytest.mark.gpu
def test_fused_latent_concat_forward_backward_matches_torch():
    """_LatentConcatFn (tmjx_latent_concat + tmjx_latent_concat_bwd) inside IntentionPolicy.forward against the torch composition
    chunk / exp / mul / add / cat: outputs and parameter gradients."""
    from track_mjx_amd.agent.networks import IntentionPolicy
    torch.manual_seed(0)
    pol = IntentionPolicy(696, 470, 38, 60, (64, 64), (64, 64)).to(DEV)
    obs = torch.randn(20, 128, 696, device=DEV)                 # 2560 rows: the fused path
    eps = torch.randn(20, 128, 60, device=DEV)
    up = torch.randn(20, 128, 76, device=DEV)
    logits, fc2 = pol(obs, eps=eps, return_fc2=True)
    g = torch.autograd.grad((logits * up).sum() + fc2.square().sum(), list(pol.parameters()))
    # reference: the same network on small chunks (below the fused path's row threshold -> torch ops)
    ref_logits, ref_fc2 = [], []
    for t in range(20):
        for c in range(0, 128, 8):
            l, f = pol(obs[t, c:c + 8], eps=eps[t, c:c + 8], return_fc2=True)
            ref_logits.append(l); ref_fc2.append(f)
    rl = torch.stack(ref_logits).view(20, 16, 8, 76).reshape(20, 128, 76)
    rf = torch.stack(ref_fc2).view(20, 16, 8, 120).reshape(20, 128, 120)
    assert float((logits - rl).abs().max()) < 2e-5 and float((fc2 - rf).abs().max()) < 2e-5
    gr = torch.autograd.grad((rl * up).sum() + rf.square().sum(), list(pol.parameters()))
    for a, b in zip(g, gr):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max() + 1e-6)
    # the second gradient of fc2 handed to the backward kernel (tmjx_latent_concat_bwd_add, what the learner does with the KL term's) instead of
    # seeding autograd with it: the same sum, bit for bit
    logits, fc2 = pol(obs, eps=eps, return_fc2=True)
    g2 = 2 * fc2.detach()
    g_seed = torch.autograd.grad([logits, fc2], list(pol.parameters()), grad_outputs=[up, g2])
    logits, fc2 = pol(obs, eps=eps, return_fc2=True)
    h = pol.latent_grad_handle
    assert h is not None and h.matches(fc2)
    h.add(g2.view(-1, 120))
    g_add = torch.autograd.grad([logits], list(pol.parameters()), grad_outputs=[up])
    assert not h.pending
    for a, b in zip(g_add, g_seed):
        assert torch.equal(a, b)
    # a parked gradient that no backward pass consumed must not be dropped silently: the next forward raises
    logits, fc2 = pol(obs, eps=eps, return_fc2=True)
    pol.latent_grad_handle.add(g2.view(-1, 120))
    with pytest.raises(RuntimeError, match="never consumed"):
        pol(obs, eps=eps, return_fc2=True)
    pol.latent_grad_handle = None
    # two policies in one process do not share the slot
    pol2 = type(pol)(696, 470, 38, 60, (64, 64), (64, 64)).to(DEV)
    lo1, f1 = pol(obs, eps=eps, return_fc2=True)
    lo2, f2 = pol2(obs, eps=eps, return_fc2=True)
    assert pol.latent_grad_handle is not pol2.latent_grad_handle and pol.latent_grad_handle.matches(f1) and pol2.latent_grad_handle.matches(f2)
    l8, _ = pol(obs[0, :8], eps=eps[0, :8], return_fc2=True)      # the unfused path (few rows) leaves no handle behind
    assert pol.latent_grad_handle is None


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,kmajor", [(2048, 256, 256, False), (2048, 256, 472, True), (2048, 76, 256, False), (132, 64, 36, False),
                                          (100, 40, 20, True), (64, 33, 7, False), (2048, 512, 1024, False)])
def test_lds_free_linear_matches_torch(M, N, K, kmajor, monkeypatch):
    """tmjx_linear_nolds (the acting policy's dense layers next to the other env group's physics kernel): the matrix-core variant
    (K % 4 == 0: v_mfma_f32_16x16x4_f32 fed from global memory) and the vector variant against a float64 product; ragged M / N /
    K tails, row-major and K-major A."""
    import ctypes as C
    from track_mjx_amd import hip as _hip
    L = _hip.lib()
    A = torch.randn((K, M), device=DEV).t() if kmajor else torch.randn((M, K), device=DEV)
    if kmajor and (M % 4 or A.stride(1) % 4):
        pytest.skip("K-major A needs M % 4 == 0")
    W, b = torch.randn((N, K), device=DEV), torch.randn(N, device=DEV)
    ref = A.double() @ W.double().t() + b.double()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for valu in (False, True):
        if valu:
            monkeypatch.setenv("TMJX_NOLDS_VALU", "1")
        out = torch.full((M, N), float("nan"), device=DEV)
        _hip.check(L.tmjx_linear_nolds(p(A), A.stride(0), A.stride(1), p(W), p(b), p(out), M, N, K, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nolds")
        torch.cuda.synchronize()
        assert float((out.double() - ref).abs().max()) <= 5e-6 * float(ref.abs().max()), (valu, M, N, K)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,lda,norm", [(1368, 256, 472, 696, True), (1364, 256, 256, 256, False), (1365, 120, 256, 256, False), (1365, 256, 288, 288, False),
                                            (1365, 76, 256, 256, False), (2732, 1024, 472, 696, True), (64, 64, 32, 32, False), (1, 1, 4, 4, False),
                                            (130, 65, 36, 40, True), (63, 33, 8, 8, False)])
def test_lds_tiled_acting_layer_matches_torch(M, N, K, lda, norm):
    """tmjx_linear_act (round 5: the acting policy's dense layer through a 20 KB LDS tile — what a CU has free next to twelve resident physics
    workgroups): C = op(A) W^T + b against a float64 product at the acting path's shapes (an env group's rows x the 2 x 256 / rodent-mc-intention
    widths, the first layer reading 472 of the observation's 696 columns with the normaliser applied on the way) and ragged ones; without a bias;
    operands the entry does not take are refused, not mangled.  Round 6: the entry picks 32-row tiles while 64-row ones would leave CUs without a
    workgroup — the 1 365-row shapes here take those, the 2 732 x 1 024 one the 64-row tiles."""
    import ctypes as C
    from track_mjx_amd import hip as _hip
    L = _hip.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    buf = torch.randn((M, lda), generator=g, device=DEV)
    A = buf[:, :K]
    W = torch.randn((N, K), generator=g, device=DEV) / K ** 0.5
    b = torch.randn(N, generator=g, device=DEV)
    mean = torch.randn(K, generator=g, device=DEV) * 0.3 if norm else None
    istd = (torch.rand(K, generator=g, device=DEV) + 0.5) if norm else None
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.tmjx_linear_act_ok(p(A), lda, p(W), K, K) == 1
    a64 = (A.double() - mean.double()) * istd.double() if norm else A.double()
    for bias in (b, None):
        out = torch.full((M, N), float("nan"), device=DEV)
        _hip.check(L.tmjx_linear_act(p(A), lda, p(W), K, p(bias), p(out), M, N, K, p(mean), p(istd), stream), "tmjx_linear_act")
        torch.cuda.synchronize()
        ref = a64 @ W.double().t() + (bias.double() if bias is not None else 0.0)
        assert float((out.double() - ref).abs().max()) <= 5e-6 * max(float(ref.abs().max()), 1.0), (M, N, K, bias is None)
    # K not a multiple of 4 / unaligned rows: refused (the caller falls back to tmjx_linear_nolds)
    assert L.tmjx_linear_act_ok(p(A), lda, p(W), K, K - 1) == 0 and L.tmjx_linear_act_ok(p(buf[:, 1:]), lda, p(W), K, K) == 0
    assert L.tmjx_linear_act(p(buf[:, 1:]), lda, p(W), K, p(b), p(out), M, N, K, None, None, stream) == -22


@pytest.mark.gpu
def test_philox_known_answers_and_self_advancing_minibatch():
    """The device-side noise of the SGD step: (1) Philox4x32-10 against the known-answer vectors of the Random123 distribution (kat_vectors:
    zero, all-ones and pi-digit inputs); (2) tmjx_minibatch_begin: rows = the slot's slice of the permutation, bit-identical to index_select,
    the state advanced exactly once per launch, draws N(0, 1) by their first four moments, reproducible for the same (seed, counter) and
    different for the next counter."""
    import ctypes as C
    import torch
    from track_mjx_amd import hip
    from track_mjx_amd.agent import losses
    from track_mjx_amd.agent.networks import RunningStatistics
    dev = torch.device("cuda:0")
    L = hip.lib()
    kat = [([0, 0, 0, 0, 0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 6, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for inp, want in kat:
        a = torch.tensor(np.array(inp, dtype=np.uint32).view(np.int32), device=dev)
        o = torch.zeros(4, dtype=torch.int32, device=dev)
        hip.check(L.tmjx_philox4x32_10(C.c_void_p(a.data_ptr()), C.c_void_p(o.data_ptr()), None), "tmjx_philox4x32_10")
        torch.cuda.synchronize()
        assert o.cpu().numpy().view(np.uint32).tolist() == want, [hex(v) for v in o.cpu().numpy().view(np.uint32)]

    g = torch.Generator(device=dev).manual_seed(6)
    T, R, W, A, B, Z = 20, 4096, 696, 38, 1024, 60
    buf = {"observation": torch.randn(T, R, W, generator=g, device=dev), "raw_action": torch.randn(T, R, A, generator=g, device=dev),
           "log_prob": torch.randn(T, R, generator=g, device=dev), "reward": torch.randn(T, R, generator=g, device=dev),
           "discount": torch.rand(T, R, generator=g, device=dev), "truncation": torch.rand(T, R, generator=g, device=dev),
           "next_observation_last": torch.randn(R, W, generator=g, device=dev), "_B": B}
    perm = torch.randperm(R, generator=g, device=dev)
    norm = RunningStatistics(W, dev)
    norm.update(buf["observation"])
    state = torch.zeros(16 + 16 * 64, dtype=torch.long, device=dev)          # TMJX_MINIBATCH_STATE_WORDS
    state[0], state[1] = 7, 2
    d1 = losses.minibatch_begin(buf, perm, state, 1234, norm, Z)
    assert state[:2].tolist() == [8, 3] and not state[2:].any()
    idx = perm[2 * B:3 * B]
    assert torch.equal(d1["observation_normalized"], norm.normalize(buf["observation"].index_select(1, idx)))
    assert torch.equal(d1["next_observation_last_normalized"], norm.normalize(buf["next_observation_last"].index_select(0, idx)))
    for k in ("raw_action", "log_prob", "reward", "discount", "truncation"):
        assert torch.equal(d1[k], buf[k].index_select(1, idx)), k
    d2 = losses.minibatch_begin(buf, perm, state, 1234, norm, Z)                 # counter 8, slot 3
    state[0], state[1] = 7, 2
    d3 = losses.minibatch_begin(buf, perm, state, 1234, norm, Z, advance=False)  # counter 7 again
    assert state[:2].tolist() == [7, 2] and not state[2:].any()
    for k in ("latent_eps", "entropy_noise"):
        x = d1[k].double().flatten()
        n = x.numel()
        m, v = x.mean().item(), x.var().item()
        skew, kurt = ((x - m) ** 3).mean().item() / v ** 1.5, ((x - m) ** 4).mean().item() / v ** 2
        print(f"{k}: n {n} mean {m:+.4f} var {v:.4f} skew {skew:+.4f} kurt {kurt:.4f} max |x| {x.abs().max().item():.2f}")
        assert abs(m) < 5 / n ** 0.5 and abs(v - 1) < 5 * (2 / n) ** 0.5 and abs(skew) < 5 * (6 / n) ** 0.5 and abs(kurt - 3) < 5 * (24 / n) ** 0.5
        assert torch.equal(d1[k], d3[k]) and not torch.equal(d1[k], d2[k])
        assert abs(torch.corrcoef(torch.stack([d1[k].flatten(), d2[k].flatten()]))[0, 1].item()) < 5 / n ** 0.5
    assert abs(torch.corrcoef(torch.stack([d1["latent_eps"].flatten()[:100000], d1["entropy_noise"].flatten()[:100000]]))[0, 1].item()) < 0.02


@pytest.mark.gpu
def test_device_side_noise_of_the_acting_policy():
    """act_rng="device": tmjx_latent_concat / tmjx_sample_action draw their N(0, 1) noise themselves (Philox streams 2 / 3 of the seed and a
    device-side draw counter that tmjx_sample_action advances).  Checked: the standardised residuals of the latent sample and of the raw action
    are N(0, 1) by their first four moments, one inference = one counter value, the same counter reproduces the draws, the next one does not,
    and the learner's act() gives fresh actions on every call (eager and as a replayed graph)."""
    import ctypes as C
    from track_mjx_amd import hip
    dev = torch.device("cuda:0")
    L = hip.lib()
    g = torch.Generator(device=dev).manual_seed(9)
    n, Z, W, ref, A = 4096, 60, 696, 470, 38
    fc2 = torch.randn((n, 2 * Z), generator=g, device=dev) * 0.5
    obs = torch.randn((n, W), generator=g, device=dev)
    logits = torch.randn((n, 2 * A), generator=g, device=dev)
    state = torch.zeros(2, dtype=torch.long, device=dev)
    state[0] = 5
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    wd = Z + W - ref

    def concat():
        x = torch.full((n, (wd + 3) // 4 * 4), float("nan"), device=dev)
        hip.check(L.tmjx_latent_concat(p(fc2), None, p(obs), p(x), n, Z, W, ref, obs.stride(0), obs.stride(1), None, None, x.shape[1], 77, p(state), None), "concat")
        return x

    def sample():
        raw, act, lp = torch.empty((n, A), device=dev), torch.empty((A, n), device=dev), torch.empty(n, device=dev)
        hip.check(L.tmjx_sample_action(p(logits), None, p(raw), p(act), p(lp), n, A, 77, p(state), None), "sample")
        return raw, act, lp

    x1, x1b = concat(), concat()
    assert state.tolist() == [5, 0] and torch.equal(x1, x1b)
    assert torch.equal(x1[:, Z:wd], obs[:, ref:]) and not x1[:, wd:].any() and torch.isfinite(x1).all()       # copied columns, zeroed pad
    r1, a1, lp1 = sample()
    torch.cuda.synchronize()
    assert state.tolist() == [6, 0]
    x2 = concat()
    r2, _, _ = sample()
    assert state.tolist() == [7, 0] and not torch.equal(x1[:, :Z], x2[:, :Z]) and not torch.equal(r1, r2)
    state[0] = 5
    r1b, _, _ = sample()
    assert torch.equal(r1, r1b)
    eps = (x1[:, :Z] - fc2[:, :Z]) / torch.exp(0.5 * fc2[:, Z:])
    scale = torch.nn.functional.softplus(logits[:, A:]) + 0.001
    nz = (r1 - logits[:, :A]) / scale
    for name, v in (("latent eps", eps), ("action noise", nz)):
        xx = v.double().flatten()
        k = xx.numel()
        m, var = xx.mean().item(), xx.var().item()
        skew, kurt = ((xx - m) ** 3).mean().item() / var ** 1.5, ((xx - m) ** 4).mean().item() / var ** 2
        print(f"{name}: n {k} mean {m:+.4f} var {var:.4f} skew {skew:+.4f} kurt {kurt:.4f}")
        assert abs(m) < 5 / k ** 0.5 and abs(var - 1) < 5 * (2 / k) ** 0.5 + 1e-3 and abs(skew) < 5 * (6 / k) ** 0.5 and abs(kurt - 3) < 5 * (24 / k) ** 0.5
    assert torch.allclose(a1.t(), torch.tanh(r1), rtol=1e-5, atol=1e-6)
    # through the learner: eager, then as a captured graph
    from track_mjx_amd.agent import ppo
    envs = [make_env_and_oracle(num_envs=64, n_clips=4, wrappers=True, seed=k)[0] for k in range(2)]
    Ln = ppo.PPOLearner(envs, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5,
                        batch_size=32, num_minibatches=4, num_updates_per_batch=2, seed=3)
    assert Ln.act_rng == "device"
    st = envs[0].reset(torch.Generator().manual_seed(1))
    acts = [Ln._act_graphed(st.obs, 0)[1]["raw_action"].clone() for _ in range(4)]       # capture happens on the first call
    torch.cuda.synchronize()
    assert all(not torch.equal(acts[i], acts[j]) for i in range(4) for j in range(i)), "every inference must draw fresh noise"


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 3])
def test_action_repeat_of_the_episode_wrapper(R):
    """wrappers.wrap(env, action_repeat=R) (wrappers.py:21,43 -> brax EpisodeWrapper.step): the env's own step R times with the same action (an
    env that terminates in an early repeat keeps stepping, as in brax's scan), rewards summed, step counter += R, observation / done /
    truncation / metrics of the last repeat, auto-reset after the last repeat.  Checked bit for bit against R launches of the same kernels
    through an un-wrapped twin restarted from the same buffers every call, and on the first (contact-free) call against the oracle."""
    from track_mjx_amd.environment import wrap
    n, ep = 32, 7
    env1, O, cl = make_env_and_oracle(num_envs=n, wrappers=False)                # no episode end, no auto-reset
    envR, _, _ = make_env_and_oracle(num_envs=n, wrappers=False)
    envR = wrap(envR, episode_length=ep, action_repeat=R)
    O = make_oracle(envR._blob, cl, "f32")
    O.set_action_repeat(R)
    g = torch.Generator().manual_seed(11)
    clip = torch.randint(0, 4, (n,), generator=g, dtype=torch.int32); start = torch.randint(0, 30, (n,), generator=g, dtype=torch.int32)
    qn = (torch.rand((74, n), generator=g) * 2 - 1) * 1e-3; vn = (torch.rand((73, n), generator=g) * 2 - 1) * 1e-3
    qn[2] += 0.05       # start 5 cm up: the first call's R env steps are contact-free, so the values can be held against the oracle tightly
    s1 = env1.reset(g, clip, start_frame=start, qpos_noise=qn, qvel_noise=vn)
    sR = envR.reset(g, clip, start_frame=start, qpos_noise=qn, qvel_noise=vn)
    first_obs = sR.obs.clone()
    envs = O.new_envs(n)
    for e in range(n):
        O.env_reset(envs, e, int(clip[e]), int(start[e]), qn[:, e].numpy(), vn[:, e].numpy())
    L = envR.layout
    steps, prev_done = np.zeros(n, np.float32), np.zeros(n, bool)
    seen_trunc = seen_term = seen_restart = False
    for call in range(10):
        a = (torch.randn((38, n), generator=g) * (0.03 if call < 4 else 1.0)).clamp(-1, 1).to(DEV)
        env1.state_buf.copy_(envR.state_buf); env1.istate_buf.copy_(envR.istate_buf)
        total = torch.zeros(n, device=DEV)
        for r in range(R):
            s1 = env1.step(s1, a); total = total + s1.reward
        inner_done = s1.done.cpu().numpy() > 0
        sR = envR.step(sR, a); torch.cuda.synchronize()
        seen_restart |= bool((prev_done & (steps > 0)).any())
        steps = np.where(prev_done, 0, steps) + R
        over = steps >= ep
        done = inner_done | over
        assert np.array_equal(envR.state_buf[L.steps_f].cpu().numpy(), steps)
        assert np.array_equal(sR.done.cpu().numpy() > 0, done) and np.array_equal(sR.info["truncation"].cpu().numpy() > 0, over & ~inner_done)
        # (an env whose state went NaN under the violent actions reports NaN metrics on both sides and is done)
        assert torch.equal(sR.reward, total) and torch.equal(envR.metrics_buf.nan_to_num(nan=777.0), env1.metrics_buf.nan_to_num(nan=777.0))
        keep, gone = torch.from_numpy(~done).to(DEV), torch.from_numpy(done).to(DEV)
        assert torch.equal(sR.obs[keep], s1.obs[keep]) and torch.equal(sR.obs[gone], first_obs[gone])
        assert torch.equal(envR.state_buf[L.qpos:L.qpos + 259][:, keep], env1.state_buf[L.qpos:L.qpos + 259][:, keep])
        assert torch.equal(envR.state_buf[L.qpos:L.qpos + 259][:, gone], envR.state_buf[L.first_phys:L.first_phys + 259][:, gone])
        seen_trunc |= bool((over & ~inner_done).any()); seen_term |= bool(inner_done.any())
        prev_done = done
        if call == 0:
            for e in range(n):
                O.env_step(envs, e, a[:, e].cpu().numpy())
            assert np.array_equal(np.array([O.env_get(envs, e, "steps")[0] for e in range(n)]), steps)
            assert np.array_equal(np.array([O.env_get(envs, e, "done")[0] for e in range(n)]) > 0, done)
            rew_err = np.abs(sR.reward.cpu().numpy() - np.array([O.env_get(envs, e, "reward")[0] for e in range(n)])).max()
            obs_err = rel_err(sR.obs.cpu().numpy(), np.stack([O.env_get(envs, e, "obs") for e in range(n)], 0))
            print(f"action_repeat {R}: summed reward err {rew_err:.2e}, obs rel err {obs_err:.2e} against the float32 oracle")
            assert rew_err < 1e-4 * R and obs_err < 2e-4
    assert seen_trunc and seen_term and seen_restart, (seen_trunc, seen_term, seen_restart)
    with pytest.raises(NotImplementedError):
        wrap(envR, episode_length=ep, randomization_fn=lambda m: m)
    with pytest.raises(ValueError):
        wrap(envR, episode_length=ep, action_repeat=0)
