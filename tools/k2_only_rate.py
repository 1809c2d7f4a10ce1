#!/usr/bin/env python3
"""What the K3 kernels cost the roll-out: two env groups on two streams stepped (a) through tmjx_step (K2 + K3) and (b) through the physics
kernel alone (tmjx_physics_step, no reward / observation / auto-reset: states drift, timing only).  GPU box: python tools/k2_only_rate.py"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd import config as _config, hip as _hip  # noqa: E402
from track_mjx_amd.environment import wrap  # noqa: E402
from track_mjx_amd.train import build_env  # noqa: E402

dev = torch.device("cuda:0")
cfg = _config.default_config()
ngrp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 4096 // ngrp
envs = [wrap(build_env(cfg, n, dev), episode_length=195) for _ in range(ngrp)]
g = torch.Generator().manual_seed(1)
sts = [e.reset(g) for e in envs]
gen = torch.Generator(device=dev).manual_seed(5)
acts = [torch.randn((38, n), generator=gen, device=dev).clamp(-1, 1) * 0.3 for _ in envs]
streams = [torch.cuda.Stream(device=dev) for _ in envs]
L = _hip.lib()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
for mode in ("step", "physics"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            for k, e in enumerate(envs):
                with torch.cuda.stream(streams[k]):
                    if mode == "step":
                        sts[k] = e.step(sts[k], acts[k])
                    else:
                        _hip.check(L.tmjx_physics_step(e._handle, p(e.state_buf), p(acts[k]), p(e.workspace), n, C.c_void_p(streams[k].cuda_stream)), "physics")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode:8s}: {4096 * 40 / dt / 1e6:.3f} M env-steps/s ({dt / 40 * 1e3:.3f} ms per control step of 4096 envs)")
