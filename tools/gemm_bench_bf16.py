#!/usr/bin/env python3
"""Time the bf16-operand GEMM kernels (tmjx_bgemm_nt forward / input gradient, tmjx_bgemm_dw) at the shapes of a config-5 minibatch step
(40 960 rows, rodent-mc-intention widths), next to torch's library bf16 GEMMs INCLUDING their operand casts (the path these kernels replaced:
a yardstick, not a product path).  Prints time, TFLOP/s and the algorithmic GB/s.  Run on the GPU box: python tools/gemm_bench_bf16.py [f32|bf16]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd.agent.networks import Bf16Shadows, _dense, bgemm_dw, bgemm_nt  # noqa: E402

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


def main():
    act = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 40960
    eb = 2 if act == torch.bfloat16 else 4
    layers = [(1024, 470, 696), (512, 1024, 1024), (512, 512, 512), (120, 512, 512), (512, 286, 288), (256, 512, 512), (256, 256, 256), (76, 256, 256), (512, 696, 696)]
    tot = {"fwd": [0, 0], "dx": [0, 0], "dw": [0, 0]}
    flops_all = 0
    for N, K, ld in layers:
        lin = _dense(K, N).to(DEV)
        sh = Bf16Shadows([lin])
        sh.refresh()
        xb = torch.randn((M, ld), device=DEV).to(act)
        x = xb[:, :K]
        dy = torch.randn((M, (N + 7) // 8 * 8), device=DEV).to(act)[:, :N]
        w = lin.weight.detach()
        fl = 2.0 * M * N * K
        x32, dy32 = x.float().contiguous(), dy.float().contiguous()
        r = {"fwd": (timeit(lambda: bgemm_nt(x, sh.w[lin], N, K, lin.bias)), timeit(lambda: torch.mm(x32.to(torch.bfloat16), w.to(torch.bfloat16).t(), out_dtype=torch.float32)),
                     (M * K * eb + M * N * 4)),
             "dx": (timeit(lambda: bgemm_nt(dy, sh.wt[lin], K, N)), timeit(lambda: torch.mm(dy32.to(torch.bfloat16), w.to(torch.bfloat16), out_dtype=torch.float32)),
                    (M * N * eb + M * K * 4)),
             "dw": (timeit(lambda: bgemm_dw(dy, x, True)), timeit(lambda: (torch.mm(dy32.to(torch.bfloat16).t(), x32.to(torch.bfloat16), out_dtype=torch.float32), dy32.sum(0))),
                    (M * N * eb + M * K * eb))}
        print(f"N={N:5d} K={K:5d}: " + "  ".join(f"{k} {a * 1e6:7.1f} us {fl / a / 1e12:6.1f} TF {by / a / 1e9:6.0f} GB/s (torch {t * 1e6:7.1f} us)" for k, (a, t, by) in r.items()), flush=True)
        for k, (a, t, _) in r.items():
            tot[k][0] += a; tot[k][1] += t
        flops_all += fl
    for k, (a, t) in tot.items():
        print(f"sum {k}: ours {a * 1e6:8.1f} us ({flops_all / a / 1e12:6.1f} TF)   torch incl. casts {t * 1e6:8.1f} us ({flops_all / t / 1e12:6.1f} TF)")


if __name__ == "__main__":
    main()
