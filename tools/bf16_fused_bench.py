#!/usr/bin/env python3
"""Time the FUSED forms of the bf16 GEMM kernels in isolation, next to the plain kernel of the same shape (config 5: 40 960 rows): Dense -> SiLU -> LayerNorm forward
(tmjx_bgemm_ln_fwd), Dense -> SiLU forward, and the consumer's input-gradient GEMM with the block's backward in its epilogue (tmjx_bgemm_ln_bwd / _silu_bwd).
What an epilogue costs beyond the GEMM it rides on.  Run on the GPU box: python tools/bf16_fused_bench.py [rows]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd.agent import networks as nw  # noqa: E402

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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
    for N, K in [(512, 512), (256, 512), (256, 256), (512, 1024), (512, 696)]:
        lin = nw._dense(K, N).to(DEV)
        cons = nw._dense(N, N).to(DEV)           # the block's consumer (N -> N) for the backward forms
        sh = nw.Bf16Shadows([lin, cons])
        sh.refresh()
        x = torch.randn((M, (K + 7) // 8 * 8), device=DEV).to(torch.bfloat16)[:, :K]
        gamma, beta = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
        dyc = torch.randn((M, N), device=DEV).to(torch.bfloat16)
        z, y, stats = nw.bgemm_ln_fwd(x, sh.w[lin], N, K, lin.bias, gamma, beta, 1e-6)
        r = {"plain nt (f32 out)": timeit(lambda: nw.bgemm_nt(x, sh.w[lin], N, K, lin.bias)),
             "ln_fwd": timeit(lambda: nw.bgemm_ln_fwd(x, sh.w[lin], N, K, lin.bias, gamma, beta, 1e-6)),
             "silu_fwd": timeit(lambda: nw.bgemm_silu_fwd(x, sh.w[lin], N, K, lin.bias)),
             "plain dx N x N (f32 out)": timeit(lambda: nw.bgemm_nt(dyc, sh.wt[cons], N, N)),
             "ln_bwd (dx of consumer N x N)": timeit(lambda: nw.bgemm_ln_bwd(dyc, sh.wt[cons], N, N, z, lin.bias, gamma, stats)),
             "silu_bwd (same)": timeit(lambda: nw.bgemm_silu_bwd(dyc, sh.wt[cons], N, N, z, lin.bias)),
             "dw": timeit(lambda: nw.bgemm_dw(dyc, x, False))}
        fl = 2.0 * M * N * K
        print(f"rows {M} N={N} K={K} ({fl / 1e9:.1f} GF): " + "  ".join(f"{k} {v:6.1f} us" for k, v in r.items()), flush=True)


if __name__ == "__main__":
    main()
