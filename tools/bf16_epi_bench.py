#!/usr/bin/env python3
"""Isolated timings of the fused-epilogue bf16 kernels at config-5 shapes (M = 40 960): plain GEMM vs + LayerNorm forward / backward epilogue vs
+ SiLU forward / backward epilogue, with the algorithmic bytes each moves.  python tools/bf16_epi_bench.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd.agent.networks import Bf16Shadows, _Block, _dense, bgemm_ln_bwd, bgemm_ln_fwd, bgemm_nt, bgemm_silu_bwd, bgemm_silu_fwd  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


M = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
for N, K in ((512, 512), (256, 512), (512, 1024), (256, 256)):
    blk = _Block(K, N).to(DEV)
    cons = _dense(N, K).to(DEV)
    sh = Bf16Shadows([blk.dense, cons]); sh.refresh()
    x16 = torch.randn((M, K), device=DEV).to(torch.bfloat16)
    dy16 = torch.randn((M, K), device=DEV).to(torch.bfloat16)
    z, y, stats = bgemm_ln_fwd(x16, sh.w[blk.dense], N, K, blk.dense.bias, blk.norm.weight, blk.norm.bias, 1e-6)
    t = {"plain fwd (f32 out)": (timeit(lambda: bgemm_nt(x16, sh.w[blk.dense], N, K, blk.dense.bias)), M * K * 2 + M * N * 4),
         "ln fwd (z f32 + y bf16)": (timeit(lambda: bgemm_ln_fwd(x16, sh.w[blk.dense], N, K, blk.dense.bias, blk.norm.weight, blk.norm.bias, 1e-6)), M * K * 2 + M * N * 6),
         "silu fwd (z f32 + y bf16)": (timeit(lambda: bgemm_silu_fwd(x16, sh.w[blk.dense], N, K, blk.dense.bias)), M * K * 2 + M * N * 6),
         "ln bwd (z in, dz bf16)": (timeit(lambda: bgemm_ln_bwd(dy16, sh.wt[cons], N, K, z, blk.dense.bias, blk.norm.weight, stats)), M * K * 2 + M * N * 6),
         "silu bwd (z in, dz bf16)": (timeit(lambda: bgemm_silu_bwd(dy16, sh.wt[cons], N, K, z, blk.dense.bias)), M * K * 2 + M * N * 6)}
    print(f"N={N} K={K}: " + "  ".join(f"{k} {a * 1e6:6.1f} us {by / a / 1e9:5.0f} GB/s" for k, (a, by) in t.items()), flush=True)
