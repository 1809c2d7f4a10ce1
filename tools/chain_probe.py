#!/usr/bin/env python3
"""Marginal costs inside the forward chain kernel: t(n hidden layers) for n = 1..4 at K0 = 256 (first layer from global memory, the others from the Y image),
with / without the last layer, both epilogue kinds."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.test_gpu_chain import _net  # noqa: E402
from tools.chain_bench import timed  # noqa: E402
from track_mjx_amd.agent.networks import chain_fwd  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for M in [int(a) for a in sys.argv[1:]] or [20480, 5120]:
    x2 = torch.randn((M, 256), generator=g, device=DEV)
    for kind in ("ln", "silu"):
        for Nf in (0, 120):
            row = []
            for nh in (1, 2, 3, 4):
                hidden, final = _net(g, 256, nh, Nf, kind)
                row.append(timed(lambda: chain_fwd(x2, hidden, final, kind)))
            print(f"rows {M} {kind:4s} last layer {Nf:3d}: " + "  ".join(f"nh={i + 1}: {t:6.1f} us" for i, t in enumerate(row)), flush=True)
