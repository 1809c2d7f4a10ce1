import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.common import make_env_and_oracle, make_oracle, rel_err
n = 32
os.environ["TMJX_IMPL"] = "wave"
envw, O, cl = make_env_and_oracle(num_envs=n, wrappers=False)
os.environ["TMJX_IMPL"] = "lane"
envl, _, _ = make_env_and_oracle(num_envs=n, wrappers=False)
O64 = make_oracle(envw._blob, cl, "f64")
rng = np.random.default_rng(0)
qpos = np.zeros((n, 74)); qvel = rng.uniform(-1e-3, 1e-3, size=(n, 73))
for e in range(n):
    c, f = e % 4, (7 * e) % 44
    qpos[e] = np.concatenate([cl.position[c, f], cl.quaternion[c, f], cl.joints[c, f]]) + rng.uniform(-1e-3, 1e-3, 74)
    qpos[e, 2] -= 0.012 * (e % 5)
act = rng.uniform(-0.1, 0.1, size=(n, 38))
for env in (envw, envl):
    env.rows("qpos").copy_(torch.from_numpy(qpos.T.astype(np.float32))); env.rows("qvel").copy_(torch.from_numpy(qvel.T.astype(np.float32)))
    env.rows("act").copy_(torch.from_numpy(act.T.astype(np.float32)))
    env.forward()
torch.cuda.synchronize()
ds = []
for e in range(n):
    d = O64.new_data(qpos[e], qvel[e]); O64.set(d, "act", act[e]); O64.forward(d); ds.append(d)
for name in ("xpos", "subtree_com", "cdof", "con_dist", "con_frame", "qfrc_actuator", "qfrc_smooth", "efc_D", "efc_aref", "qacc_smooth", "qacc", "efc_force", "qfrc_constraint", "qM"):
    gw = envw.rows(name).cpu().numpy().astype(np.float64); gl = envl.rows(name).cpu().numpy().astype(np.float64)
    if name == "subtree_com":
        ref = np.stack([O64.get(d, name).reshape(68, 3)[2] for d in ds], 1)
    elif name == "qM":
        ref = gl
    else:
        ref = np.stack([O64.get(d, name) for d in ds], 1)
    print(f"{name:16s} wave-vs-oracle {rel_err(gw, ref):.2e}  lane-vs-oracle {rel_err(gl, ref):.2e}  wave-vs-lane {rel_err(gw, gl):.2e}")
    if name == "subtree_com":
        print("   com wave", gw[:, 0], "ref", ref[:, 0])
    if name == "cdof":
        bad = np.abs(gw - ref).max(1)
        print("   worst cdof rows (dof, comp):", [(i // 6, i % 6) for i in np.argsort(-bad)[:12]], bad.max())
