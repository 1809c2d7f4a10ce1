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
import signal
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


def _set_pdeathsig():
    """preexec of the child: own session (= own process group: one killpg reaches torchrun AND its ranks) and PR_SET_PDEATHSIG so that the
    launcher dies with this process even when this process is SIGKILLed (no handler runs then)."""
    os.setsid()
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)      # PR_SET_PDEATHSIG = 1
    except Exception:  # noqa: BLE001 — not Linux / no libc: the signal handlers below still cover SIGTERM and SIGINT
        pass


def run_in_own_group(cmd: Sequence[str], env: dict) -> int:
    """Run `cmd` as the leader of a new process group and wait for it.  SIGTERM / SIGINT / SIGHUP received by this process are answered by
    killing the whole group (torchrun and every rank: orphaned ranks would keep holding the GPUs), and so is every other way out of the
    wait (an exception, KeyboardInterrupt)."""
    proc = subprocess.Popen(list(cmd), env=env, preexec_fn=_set_pdeathsig)      # stdout / stderr inherited

    def kill_group(sig=signal.SIGTERM):
        try:
            os.killpg(proc.pid, sig)
        except ProcessLookupError:
            pass

    def on_signal(signum, _frame):
        kill_group(signal.SIGTERM)
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            kill_group(signal.SIGKILL)
        os._exit(128 + signum)

    old = {}
    for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old[s] = signal.signal(s, on_signal)
        except ValueError:          # not the main thread: the finally clause below is the cover
            pass
    try:
        return int(proc.wait())
    finally:
        if proc.poll() is None:     # leaving with the child alive (exception in the wait)
            kill_group(signal.SIGTERM)
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                kill_group(signal.SIGKILL)
        else:                       # torchrun has exited: no rank of its group may outlive it
            kill_group(signal.SIGKILL)
        for s, h in old.items():
            signal.signal(s, h)


def spawn_ranks(n_gpus: int, target: Sequence[str], args: Sequence[str], runner: Callable | None = None, env: dict | None = None) -> int:
    """Start the ranks as a child process and wait for them; returns the exit code.  `runner(cmd, env=...) -> int` is the process
    starter (tests pass a recorder); the default is `run_in_own_group`."""
    cmd = rank_command(n_gpus, target, args)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    child_env.setdefault("OMP_NUM_THREADS", "4")
    if runner is None:
        runner = run_in_own_group
    print(f"[launch] {n_gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return int(runner(cmd, env=child_env))
