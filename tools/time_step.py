#!/usr/bin/env python3
"""Time tmjx_step (K2+K3 fused launch) at a given env count; prints env-steps/s. Used for kernel tuning."""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.common import make_env_and_oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--blocks", type=str, default="")
    args = ap.parse_args()
    for blk in ([int(b) for b in args.blocks.split(",")] if args.blocks else [None]):
        if blk:
            os.environ["TMJX_BLOCK"] = str(blk)
        env, _, _ = make_env_and_oracle(num_envs=args.envs, n_clips=64, wrappers=True)
        g = torch.Generator().manual_seed(0)
        st = env.reset(g)
        acts = [(torch.randn((38, args.envs), generator=g) * args.scale).clamp(-1, 1).cuda() for _ in range(4)]
        for i in range(2):
            st = env.step(st, acts[i % 4])
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(args.steps):
            st = env.step(st, acts[i % 4])
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.steps
        print(f"block={blk} envs={args.envs} ms/step={dt * 1e3:.2f} env-steps/s={args.envs / dt:.0f} done_frac={st.done.mean().item():.3f}", flush=True)
        del env


if __name__ == "__main__":
    main()
