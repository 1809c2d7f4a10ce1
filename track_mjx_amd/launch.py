"""One process per GPU, started by the program itself.

The reference runs all local devices from ONE process (`jax.pmap`, track_mjx/agent/mlp_ppo/ppo.py:409,477-480), so a user
types `python -m track_mjx.train` (or a driver `python bench.py --gpus 8`) and gets every GPU of the node.  Here the unit is one
process per GPU over RCCL, so the same command line has to start its own ranks: `spawn_ranks` runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <target> <args>` as a CHILD
process (never an exec: a process that has touched the GPU must not be replaced, and this one may be running under a profiler),
with stdout / stderr inherited — rank 0's lines are the program's lines — and returns the child's exit code.

It must be called before anything initialises the GPU in this process (the caller checks `needs_spawn` first thing in `main`).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, Sequence


def needs_spawn(n_gpus: int, environ=None) -> bool:
    """True when `n_gpus` > 1 ranks were asked for and this process is not already one of a launcher's ranks.  TMJX_FORCE_SPAWN=1 takes the
    launcher path for one rank too (the self-launch + RCCL path on a single-GPU box)."""
    env = os.environ if environ is None else environ
    return (n_gpus > 1 or (n_gpus == 1 and bool(env.get("TMJX_FORCE_SPAWN")))) and "RANK" not in env and "WORLD_SIZE" not in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_command(n_gpus: int, target: Sequence[str], args: Sequence[str], port: int | None = None) -> list[str]:
    """The child command line.  `target` = ["bench.py"] (a script) or ["-m", "track_mjx_amd.train"] (a module)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port() if port is None else port), *target, *args]


def spawn_ranks(n_gpus: int, target: Sequence[str], args: Sequence[str], runner: Callable | None = None, env: dict | None = None) -> int:
    """Start the ranks as a child process and wait for them; returns the exit code.  `runner(cmd, env=...) -> int` is the process
    starter (tests pass a recorder)."""
    cmd = rank_command(n_gpus, target, args)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    child_env.setdefault("OMP_NUM_THREADS", "4")
    if runner is None:
        runner = lambda c, env: subprocess.call(c, env=env)      # noqa: E731  (stdout / stderr inherited)
    print(f"[launch] {n_gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return int(runner(cmd, env=child_env))
