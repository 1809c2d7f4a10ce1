#!/usr/bin/env python3
"""Bit-for-bit A/B of library builds ON THE GPU: env.step of a few hundred envs, free-running from the same reset with the same actions (three action
scales, through contact and auto-reset), every state row / observation / reward compared after every control step.  The host emulation's A/B
(tests/diagnostics/kernel_ab.py) proves that a change leaves the ALGORITHM alone; this one shows whether the compiled kernels still produce the same
bits (FMA contraction and instruction selection are the compiler's).  GPU box:  python tools/gpu_bits_ab.py libA.so libB.so [libC.so ...]
Each library runs in its own process (TMJX_SO is read once); the first one is the reference."""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def child(out: str):
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    from tests.common import make_env_and_oracle
    n = 384
    env, _, _ = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
    rows = []
    for scale in (0.03, 0.3, 1.0):
        g = torch.Generator().manual_seed(7)
        st = env.reset(g)
        for step in range(24):
            a = (torch.randn((38, n), generator=g) * scale).clamp(-1, 1).cuda()
            st = env.step(st, a)
            torch.cuda.synchronize()
            rows.append(np.concatenate([env.state_buf.detach().float().cpu().numpy().ravel(), st.obs.detach().float().cpu().numpy().ravel(),
                                        st.reward.detach().float().cpu().numpy().ravel(), st.done.detach().float().cpu().numpy().ravel()]).view(np.uint32))
    np.save(out, np.stack(rows))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    import numpy as np
    libs = sys.argv[1:]
    outs = []
    for k, so in enumerate(libs):
        out = f"/tmp/bits_{k}.npy"
        subprocess.run([sys.executable, __file__, "--child", out], check=True, env={**os.environ, "TMJX_SO": so})
        outs.append(np.load(out))
    ref = outs[0]
    print(f"reference {libs[0]}: {ref.shape[0]} control steps x {ref.shape[1]} words, sha1 {hashlib.sha1(ref.tobytes()).hexdigest()[:12]}")
    for so, x in zip(libs[1:], outs[1:]):
        neq = (x != ref)
        first = int(np.argmax(neq.any(1))) if neq.any() else -1
        print(f"{so}: {'IDENTICAL' if not neq.any() else f'{int(neq.sum())} words differ, first in control step {first}'}")
        if neq.any():      # which envs: the state block is [rows][n] (env = word % n); non-finite words of the reference in those envs at that step
            n = 384
            srows = (ref.shape[1] - 0) // n
            blk = neq[:, : (ref.shape[1] // n) * n].reshape(neq.shape[0], -1, n)
            envs = sorted(set(np.nonzero(blk.any((0, 1)))[0].tolist()))
            print(f"    envs (word % {n}) with a differing word: {envs[:12]}{' ...' if len(envs) > 12 else ''} ({len(envs)} of {n})")
