#!/usr/bin/env python3
"""Where a workgroup of k_bgemm_nt spends its cycles: s_memtime stamps (100 MHz ticks? no: shader cycles) of wave 0 around prologue / K loop / epilogue stores.
python tools/bf16_stamps.py M N K [f32|bf16]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
act = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
buf = torch.zeros(((M + 79) // 80) * ((N + 511) // 512 + 4) * 4, device="cuda:0")
os.environ["TMJX_BG_STAMPS"] = hex(buf.data_ptr())
from track_mjx_amd.agent.networks import Bf16Shadows, _dense, bgemm_nt  # noqa: E402

lin = _dense(K, N).cuda()
sh = Bf16Shadows([lin]); sh.refresh()
x = torch.randn((M, K), device="cuda:0").to(act)
for _ in range(3):
    bgemm_nt(x, sh.w[lin], N, K, lin.bias)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bgemm_nt(x, sh.w[lin], N, K, lin.bias); e1.record()
torch.cuda.synchronize()
nwg = ((M + 79) // 80) * ((N + (127 if N <= 128 else 255 if N <= 256 else 511)) // (128 if N <= 128 else 256 if N <= 256 else 512))
s = buf[:nwg * 4].view(nwg, 4).cpu()
t0 = s[:, 0]
start = (t0 - t0.min()) % (1 << 24)
print(f"M={M} N={N} K={K} {act}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups")
print(f"  start spread (ticks after the first workgroup): median {start.median():.0f}, max {start.max():.0f}")
for i, nm in enumerate(("prologue", "K loop", "epilogue stores (until accepted)"), 1):
    print(f"  {nm:34s} median {s[:, i].median():9.0f}  p10 {s[:, i].quantile(0.1):9.0f}  p90 {s[:, i].quantile(0.9):9.0f} ticks")
print(f"  total per workgroup median {(s[:, 1] + s[:, 2] + s[:, 3]).median():.0f} ticks;  K tiles {K // 64}")
