#!/usr/bin/env python3
"""Time the learner's MFMA GEMM kernels (tmjx_gemm_nt / _nn / _dw) at the shapes of a PPO minibatch step, next to torch's library
GEMMs on the same data (a yardstick for tuning, not the product path).  Run on the GPU box: python tools/gemm_bench.py [cfg2|cfg4] [rows]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd.agent.networks import gemm_dw, gemm_nn, gemm_nt  # noqa: E402

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
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 20480          # rows: 20 480 (cfg2 / cfg4), 5 120 (cfg3: one rank's share of batch_size 2048 on 8 GPUs)
    print(f"M = {M}")
    if cfg == "cfg2":
        layers = [(256, 470, 696), (256, 256, 256), (120, 256, 256), (256, 286, 288), (256, 256, 256), (76, 256, 256), (256, 696, 696), (256, 256, 256), (1, 256, 256)]
    else:
        layers = [(1024, 470, 696), (512, 1024, 1024), (512, 512, 512), (120, 512, 512), (512, 286, 288), (256, 512, 512), (76, 256, 256), (512, 696, 696), (1, 256, 256)]
    tot = {"nt": [0, 0], "nn": [0, 0], "dw": [0, 0]}
    flops_all = 0
    for N, K, ld in layers:
        xb = torch.randn((M, ld), device=DEV)
        x = xb[:, :K]
        w = torch.randn((N, (K + 3) // 4 * 4), device=DEV)[:, :K]       # rows 16-byte aligned, as in the learner's flat parameter buffer
        wc = w.contiguous()                                              # (the library's operand)
        b = torch.randn(N, device=DEV)
        dy = torch.randn((M, N), device=DEV)
        fl = 2.0 * M * N * K
        xc = x.contiguous()
        r = {"nt": (timeit(lambda: gemm_nt(x, w, b)), timeit(lambda: torch.addmm(b, xc, wc.t()))),
             "nn": (timeit(lambda: gemm_nn(dy, w)), timeit(lambda: dy @ wc)),
             "dw": (timeit(lambda: gemm_dw(dy, x, True)), timeit(lambda: (dy.t() @ xc, dy.sum(0))))}
        print(f"N={N:5d} K={K:5d}: " + "  ".join(f"{k} {a * 1e6:7.1f} us {fl / a / 1e12:6.1f} TF (torch {t * 1e6:7.1f} us {fl / t / 1e12:6.1f} TF)" for k, (a, t) in r.items()), flush=True)
        for k, (a, t) in r.items():
            tot[k][0] += a; tot[k][1] += t
        flops_all += fl
    for k, (a, t) in tot.items():
        print(f"sum {k}: ours {a * 1e6:8.1f} us ({flops_all / a / 1e12:6.1f} TF)   torch {t * 1e6:8.1f} us ({flops_all / t / 1e12:6.1f} TF)")


if __name__ == "__main__":
    main()
