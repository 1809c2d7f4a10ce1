import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.common import make_env_and_oracle
n = 4096
res = {}
for impl in ("wave", "wave", "lane"):
    os.environ["TMJX_IMPL"] = impl
    env, O, cl = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
    g = torch.Generator().manual_seed(11)
    st = env.reset(g)
    outs = [env.state_buf.clone()]
    for s in range(3):
        a = (torch.randn((38, n), generator=g) * 0.5).clamp(-1, 1).cuda()
        st = env.step(st, a)
        torch.cuda.synchronize()
        outs.append(env.state_buf.clone())
    res.setdefault(impl, []).append(outs)
w0, w1, l0 = res["wave"][0], res["wave"][1], res["lane"][0]
for s in range(4):
    a, b, c = w0[s], w1[s], l0[s]
    nan_w = (~torch.isfinite(a[:259])).any(0).sum().item(); nan_l = (~torch.isfinite(c[:259])).any(0).sum().item()
    same = (a == b) | (torch.isnan(a) & torch.isnan(b))
    diff_rows = (~same).any(1).nonzero().flatten().tolist()
    qd = (a[:74] - c[:74]).abs().max(0).values
    print(f"step {s}: nan envs wave {nan_w} lane {nan_l}; wave run-to-run differing rows {diff_rows[:10]} (n={len(diff_rows)}), envs {(~same).any(0).sum().item()}; wave-vs-lane qpos median {qd.median().item():.2e} max {qd[torch.isfinite(qd)].max().item():.2e}")
