#!/usr/bin/env python3
"""Whole-chain kernels (tmjx_chain_fwd / tmjx_chain_bwd) against the layer-by-layer launches they replace, isolated, on the 2x256 nets' chains:
microseconds per chain (HIP events over `--iters` back-to-back calls on one stream).  usage: python tools/chain_bench.py [rows ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.test_gpu_chain import _bwd_layer_by_layer, _layer_by_layer, _net  # noqa: E402
from track_mjx_amd.agent.networks import chain_bwd, chain_fwd  # noqa: E402

DEV = "cuda:0"
ITERS = 50


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e3


def main():
    rows = [int(a) for a in sys.argv[1:]] or [20480, 5120, 1365]
    g = torch.Generator(device=DEV).manual_seed(0)
    for M in rows:
        for name, K0, lda, Nf, kind in (("encoder + fc2", 470, 696, 120, "ln"), ("decoder + head", 286, 288, 76, "ln"), ("critic + head", 696, 696, 1, "silu")):
            x2 = torch.randn((M, lda), generator=g, device=DEV)[:, :K0]
            hidden, final = _net(g, K0, 2, Nf, kind)
            t_chain = timed(lambda: chain_fwd(x2, hidden, final, kind))
            t_layers = timed(lambda: _layer_by_layer(x2, hidden, final, kind))
            flops = 2.0 * M * (K0 * 256 + 256 * 256 + 256 * Nf)
            print(f"forward  {name:15s} rows {M:6d}: chain {t_chain:7.1f} us ({flops / t_chain / 1e6:6.1f} TF/s)   layer by layer {t_layers:7.1f} us ({flops / t_layers / 1e6:6.1f} TF/s)", flush=True)
            saved, _ = chain_fwd(x2, hidden, final, kind)
            gr = torch.randn((M,) if Nf == 1 else (M, Nf), generator=g, device=DEV)
            if kind == "ln":
                blocks = [(hidden[l][0], saved[l][0], hidden[l][1], hidden[l][2], saved[l][2]) for l in (1, 0)]
            else:
                blocks = [(hidden[l][0], saved[l][0], hidden[l][1]) for l in (1, 0)]
            w0, cols = (hidden[0][0], 60) if K0 == 286 else (None, None)
            t_chain = timed(lambda: chain_bwd(gr, final[0], blocks, kind, w0, cols))
            t_layers = timed(lambda: _bwd_layer_by_layer(gr, final[0], blocks, kind, w0, cols))
            flops = 2.0 * M * (256 * Nf * (Nf > 1) + 256 * 256 + (256 * 64 if cols else 0))
            print(f"backward {name:15s} rows {M:6d}: chain {t_chain:7.1f} us ({flops / t_chain / 1e6:6.1f} TF/s)   layer by layer {t_layers:7.1f} us ({flops / t_layers / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
