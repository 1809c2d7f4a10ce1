#!/usr/bin/env python3
"""In-kernel clock stamps of the forward chain kernel (tmjx_chain_fwd_t.prof): median over the workgroups of the cycles between consecutive stamps
(kernel start | K loop of layer 0 | epilogue 0 | K loop 1 | epilogue 1 | ... | last layer's K loop | its epilogue).  usage: python tools/chain_stamps.py [rows]"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.test_gpu_chain import _net  # noqa: E402
from track_mjx_amd.agent.networks import _launch, chain_bwd, chain_fwd_desc  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
g = torch.Generator(device=DEV).manual_seed(0)
for name, K0, lda, Nf, kind in (("encoder + fc2", 470, 696, 120, "ln"), ("decoder + head", 286, 288, 76, "ln"), ("critic + head", 696, 696, 1, "silu")):
    x2 = torch.randn((M, lda), generator=g, device=DEV)[:, :K0]
    hidden, final = _net(g, K0, 2, Nf, kind)
    d, saved, out = chain_fwd_desc(x2, hidden, final, kind)
    prof = torch.zeros((4096, 16), dtype=torch.int64, device=DEV)
    d.prof = prof.data_ptr()
    for _ in range(3):
        _launch("tmjx_chain_fwd", DEV, C.byref(d))
    torch.cuda.synchronize()
    p = prof.cpu()
    n = int((p[:, 0] != 0).sum())
    p = p[:n]
    k = int((p[0] != 0).sum())
    dt = (p[:, 1:k] - p[:, :k - 1]).float()
    print(f"forward  {name} rows {M}: {n} workgroups; cycles between stamps (median): " + " | ".join(f"{int(v)}" for v in dt.median(0).values) + f"  total {int((p[:, k - 1] - p[:, 0]).float().median())}", flush=True)
    # backward: kernel start | first GEMM (from G) | epilogue of the last block | GEMM | epilogue | ... | trailing dx GEMM | its stores
    gr = torch.randn((M,) if Nf == 1 else (M, Nf), generator=g, device=DEV)
    if kind == "ln":
        blocks = [(hidden[l][0], saved[l][0], hidden[l][1], hidden[l][2], saved[l][2]) for l in (1, 0)]
    else:
        blocks = [(hidden[l][0], saved[l][0], hidden[l][1]) for l in (1, 0)]
    w0, cols = (hidden[0][0], 60) if K0 == 286 else (None, None)
    prof.zero_()
    for _ in range(3):
        chain_bwd(gr, final[0], blocks, kind, w0, cols, prof=prof)
    torch.cuda.synchronize()
    p = prof.cpu()
    n = int((p[:, 0] != 0).sum())
    p = p[:n]
    k = int((p[0] != 0).sum())
    dt = (p[:, 1:k] - p[:, :k - 1]).float()
    print(f"backward {name} rows {M}: {n} workgroups; cycles between stamps (median): " + " | ".join(f"{int(v)}" for v in dt.median(0).values) + f"  total {int((p[:, k - 1] - p[:, 0]).float().median())}", flush=True)
