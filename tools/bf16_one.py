#!/usr/bin/env python3
"""One bf16 GEMM shape, launched a few times (for rocprofv3 --pmc / --kernel-trace): python tools/bf16_one.py <fwd|dx|dw> M N K [f32|bf16]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd.agent.networks import Bf16Shadows, _dense, bgemm_dw, bgemm_nt  # noqa: E402

op, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
act = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else torch.float32
DEV = "cuda:0"
lin = _dense(K, N).to(DEV)
sh = Bf16Shadows([lin]); sh.refresh()
x = torch.randn((M, (K + 7) // 8 * 8), device=DEV).to(act)[:, :K]
dy = torch.randn((M, (N + 7) // 8 * 8), device=DEV).to(act)[:, :N]
fn = {"fwd": lambda: bgemm_nt(x, sh.w[lin], N, K, lin.bias), "dx": lambda: bgemm_nt(dy, sh.wt[lin], K, N), "dw": lambda: bgemm_dw(dy, x, True)}[op]
for _ in range(6):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"{op} M={M} N={N} K={K} {act}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
