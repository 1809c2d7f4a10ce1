#!/usr/bin/env python3
"""Two env objects, same reset, same actions: the state buffers must stay bit-identical (the physics kernel has no atomics and no
cross-wave communication).  GPU box: [TMJX_SO=...] python tools/determinism_check.py [n_envs] [steps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.common import make_env_and_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
envs = [make_env_and_oracle(num_envs=n, n_clips=4, wrappers=True, seed=0)[0] for _ in range(2)]
sts = [e.reset(torch.Generator().manual_seed(10)) for e in envs]
g = torch.Generator().manual_seed(3)
bad = 0
for t in range(steps):
    a = (torch.randn((38, n), generator=g) * 0.5).clamp(-1, 1).cuda()
    sts = [e.step(s, a) for e, s in zip(envs, sts)]
    torch.cuda.synchronize()
    d = (envs[0].state_buf != envs[1].state_buf) & ~(torch.isnan(envs[0].state_buf) & torch.isnan(envs[1].state_buf))      # (NaN states of blown-up envs count as equal)
    if d.any():
        rows = d.any(dim=1).nonzero().flatten().tolist()
        print(f"step {t}: DIFFER in {int(d.sum())} words, {int(d.any(dim=0).sum())} envs, first rows {rows[:8]}")
        bad += 1
        if bad >= 3:
            break
    else:
        print(f"step {t}: identical")
print("result:", "DETERMINISTIC" if bad == 0 else "NON-DETERMINISTIC")
