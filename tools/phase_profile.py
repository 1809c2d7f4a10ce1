#!/usr/bin/env python3
"""In-kernel phase profile of the wave-per-env physics kernel (s_memtime stamps; TMW_PROFILE build).
Run on the GPU box:  TMJX_SO=track_mjx_amd/libtmjx_hip_prof.so python tools/phase_profile.py"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.common import make_env_and_oracle

NAMES = ["position", "velocity+M", "factor", "invert_L", "make_constraint", "solve(qacc_smooth)", "cg init (3 cost evals + grad)",
         "cg linesearch", "cg update (+euler rhs)", "euler factor", "[count] solves with <= 16 active rows, per 1000", "euler solve", "load/store/other", "  (ls: row set-up)", "  (ls: bracket iterations)", "  (misc between sub-phases)",
         "  (J^T force)", "  (M^-1 grad)", "[count] solves with 17..32 active rows, per 1000", "  (eval_cost: J*q)", "  (vel: dof velocity scan)", "  (vel: cdof_dot + acc scan)", "  (vel: body forces)", "  (vel: subtree sums)", "  (M*x: column part)", "  (M*x: row part)", "  (M^-1: column part)", "  (pos: local transforms)", "  (pos: pointer jumping)", "  (pos: com)", "  (pos: collision)", "  (pos: cdof)",
         "  (ls: search_q = N s)", "  (ls: J*search)", "  (ls: three wave sums, tolerances)", "  (ls: points 0 and Newton)", "[count] CG iterations, per 1000 substeps", "[count] bracket iterations run, per 1000 substeps", "[count] line searches ended by the d0 == 0 shortcut, per 1000 substeps", "-"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env, _, _ = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
g = torch.Generator().manual_seed(0)
st = env.reset(g)
a = (torch.randn((38, n), generator=g) * 0.3).clamp(-1, 1).cuda()
for _ in range(3):
    st = env.step(st, a)
env.workspace.zero_()
torch.cuda.synchronize()
env.physics(a, 10)
torch.cuda.synchronize()
prof = env.workspace.flatten().view(torch.int64)[: n * 40].view(n, 40).double()
tot = prof[:, :36].sum(1) - prof[:, 10] - prof[:, 18]
print(f"envs={n} substeps=10: mean cycles/env-step {tot.mean().item():.0f} (min {tot.min().item():.0f} max {tot.max().item():.0f}); s_memtime ticks at 100 MHz")
for i, nm in enumerate(NAMES):
    print(f"  {nm:34s} {prof[:, i].mean().item() / 10:10.0f} ticks/substep  {100 * prof[:, i].mean().item() / tot.mean().item():5.1f} %")
