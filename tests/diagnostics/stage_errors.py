#!/usr/bin/env python3
"""Where does the HIP physics kernel lose accuracy against the float32 restatement of MJX's dense formulation?  Teacher-forced: every substep
starts from the float64 oracle's state; after `forward` the intermediates of the HIP kernel and of the float32 oracle are compared with the
float64 oracle's, stage by stage (per env: max |err| / max |ref| over the stage's array; the table prints the MEDIAN and the 90th percentile
over env-substeps), then the full substep's qvel / qpos.   usage: python tests/diagnostics/stage_errors.py [scale] [substeps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.common import make_env_and_oracle, make_oracle, rel_err  # noqa: E402
from tests.test_gpu_parity_strict import PHYS, _init_states, _oracle_states, _push  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 64
env, O32, cl = make_env_and_oracle(num_envs=n, wrappers=False)
O64 = make_oracle(env._blob, cl, "f64")
rng = np.random.default_rng(11)
qpos, qvel = _init_states(cl, n, rng)
d32 = [O32.new_data(qpos[e], qvel[e]) for e in range(n)]
d64 = [O64.new_data(qpos[e], qvel[e]) for e in range(n)]
par = env.walker.model["dof_parentid"]


def sparse(dense):
    out = []
    for i in range(73):
        j = i
        while j >= 0:
            out.append(dense[i * 73 + j]); j = par[j]
    return np.array(out)


stages = ["xpos", "cdof", "qM", "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "efc_D", "efc_aref", "qacc", "efc_force", "qvel", "qpos"]
errs = {s: {"hip": [], "f32": [], "hip_vs_f32": []} for s in stages}
for sub in range(nsub):
    a = np.clip(rng.normal(size=(n, 38)) * scale, -1, 1)
    st = _oracle_states(O64, d64)
    _push(env, st)
    for e in range(n):
        for k, v in st.items():
            O32.set(d32[e], k, v[:, e])
    # forward intermediates from the identical state (ctrl = this substep's action through the activation state already in `act`)
    env.forward()
    torch.cuda.synchronize()
    f32d = [O32.new_data(st["qpos"][:, e], st["qvel"][:, e]) for e in range(n)]
    f64d = [O64.new_data(st["qpos"][:, e], st["qvel"][:, e]) for e in range(n)]
    for e in range(n):
        for O, dd in ((O32, f32d), (O64, f64d)):
            O.set(dd[e], "act", st["act"][:, e]); O.set(dd[e], "qacc_warmstart", st["qacc_warmstart"][:, e]); O.forward(dd[e])
    for s in stages[:-2]:
        got = env.rows(s).cpu().numpy().astype(np.float64)
        if s == "qM":
            r64 = np.stack([sparse(O64.get(d, s)) for d in f64d], 1); r32 = np.stack([sparse(O32.get(d, s)) for d in f32d], 1)
        else:
            r64 = np.stack([O64.get(d, s) for d in f64d], 1); r32 = np.stack([O32.get(d, s) for d in f32d], 1)
        errs[s]["hip"].append(rel_err(got, r64, axis=0)); errs[s]["f32"].append(rel_err(r32, r64, axis=0)); errs[s]["hip_vs_f32"].append(rel_err(got, r32, axis=0))
    _push(env, st)
    env.physics(torch.from_numpy(a.T.astype(np.float32)).contiguous().cuda(), 1)
    torch.cuda.synchronize()
    for e in range(n):
        O32.step(d32[e], a[e]); O64.step(d64[e], a[e])
    ref = _oracle_states(O64, d64, ("qpos", "qvel")); r32 = _oracle_states(O32, d32, ("qpos", "qvel"))
    for s in ("qvel", "qpos"):
        got = env.rows(s).cpu().numpy()
        errs[s]["hip"].append(rel_err(got, ref[s], axis=0)); errs[s]["f32"].append(rel_err(r32[s], ref[s], axis=0)); errs[s]["hip_vs_f32"].append(rel_err(got, r32[s], axis=0))
print(f"action scale {scale}, {n} envs x {nsub} teacher-forced substeps; relative error per env-substep against the float64 oracle")
print(f"{'stage':16s} {'HIP median':>11s} {'f32 median':>11s} {'ratio':>6s} | {'HIP p90':>10s} {'f32 p90':>10s} {'ratio':>6s} | {'HIP vs f32 median':>18s} | within 1e-5: HIP / f32")
for s in stages:
    h, f, hf = (np.concatenate(errs[s][k]) for k in ("hip", "f32", "hip_vs_f32"))
    mh, mf = np.median(h), np.median(f)
    ph, pf = np.quantile(h, .9), np.quantile(f, .9)
    print(f"{s:16s} {mh:11.2e} {mf:11.2e} {mh / max(mf, 1e-30):6.2f} | {ph:10.2e} {pf:10.2e} {ph / max(pf, 1e-30):6.2f} | {np.median(hf):18.2e} | {np.mean(h <= 1e-5):.3f} / {np.mean(f <= 1e-5):.3f}")
