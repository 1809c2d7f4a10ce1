"""`python bench.py --gpus N` / `python -m track_mjx_amd.train num_gpus=N` start their own ranks (track_mjx_amd/launch.py).

The reference reaches every local device from one process (jax.pmap, track_mjx/agent/mlp_ppo/ppo.py:409,477-480); here the same command
line has to spawn one process per GPU — as a child, before the GPU is touched.  CPU only: the command line, the branch taken, and a
real 2-rank gloo run through the self-launch path (`--dry-run-ranks`)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from track_mjx_amd import launch  # noqa: E402


def test_needs_spawn_only_outside_a_launcher():
    assert not launch.needs_spawn(1, {})
    assert launch.needs_spawn(2, {})
    assert launch.needs_spawn(8, {"PATH": "x"})
    assert not launch.needs_spawn(8, {"RANK": "3", "WORLD_SIZE": "8"})
    assert not launch.needs_spawn(8, {"WORLD_SIZE": "8"})


def test_rank_command_is_the_drivers_own_launch_line():
    cmd = launch.rank_command(8, ["bench.py"], ["--gpus", "8", "--steps", "5"], port=29533)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29533", "bench.py", "--gpus", "8", "--steps", "5"]
    cmd = launch.rank_command(2, ["-m", "track_mjx_amd.train"], ["num_gpus=2"], port=1)
    assert cmd[-3:] == ["-m", "track_mjx_amd.train", "num_gpus=2"]


def test_bench_main_takes_the_launch_branch_and_relays_the_exit_code(monkeypatch):
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    def recorder(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"], runner=recorder)
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]          # the ranks get the caller's own arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_train_main_takes_the_launch_branch(monkeypatch):
    from track_mjx_amd import train
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}
    rc = train.main(["num_gpus=4", "max_training_steps=1"], runner=lambda cmd, env: seen.setdefault("cmd", cmd) and 0)
    assert rc == 0
    cmd = seen["cmd"]
    assert "--nproc-per-node=4" in cmd and cmd[-4:] == ["-m", "track_mjx_amd.train", "num_gpus=4", "max_training_steps=1"]


def test_self_launched_two_ranks_really_run_and_rank0_line_comes_back():
    """`python bench.py --gpus 2 --dry-run-ranks` exactly as a driver would type it (no RANK / WORLD_SIZE in the environment): the child
    torch.distributed.run starts two ranks, they meet in a gloo all-reduce, rank 0's JSON line arrives on the parent's stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run-ranks"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["asked_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["steps"] == 4
    assert "[launch] 2 ranks" in res.stderr


def test_self_launched_eight_ranks_of_the_cfg3_bench_line():
    """`python bench.py --gpus 8 --config cfg3 --dry-run-ranks`: BASELINE configs[2]'s launch — EIGHT real ranks through the self-launch path
    (the driver's own command line, track_mjx_amd/launch.py), a gloo all-reduce over all of them, and the batch arithmetic the ranks would hand
    the learner: 32 768 envs, the reference's batch_size 2048 = 256 minibatch rows per GPU, one unroll per training step
    (track_mjx/agent/mlp_ppo/ppo.py:237-239,477-480)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--config", "cfg3", "--steps", "2", "--warmup", "1", "--dry-run-ranks"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 8 and out["asked_gpus"] == 8 and c["ranks_seen"] == 8 and c["parallelism"] == "dp8"
    assert c["envs_total"] == 32768 and c["global_batch"] == 2048 and c["minibatch_rows_per_gpu"] == 256 and c["unrolls_per_training_step"] == 1
    assert "[launch] 8 ranks" in res.stderr


def test_env_group_sizes_for_the_pipelined_rollout():
    """ppo.group_sizes / default_groups: every env in exactly one group, sizes multiples of 4 (but for a remainder), at most 4 apart; three groups
    by default (bench.py --pipeline 0, train.py rollout_groups)."""
    from track_mjx_amd.agent import ppo
    assert ppo.group_sizes(4096, 3) == [1368, 1364, 1364] and ppo.group_sizes(4096, 2) == [2048, 2048] and ppo.group_sizes(8192, 3) == [2732, 2732, 2728]
    for n in (1, 2, 6, 16, 64, 130, 4095, 4096, 8192):
        for g in (1, 2, 3, 4):
            s = ppo.group_sizes(n, g)
            assert sum(s) == n and all(x > 0 for x in s) and len(s) <= g and max(s) - min(s) <= 4, (n, g, s)
    assert ppo.default_groups(4096) == 3 and ppo.default_groups(8192) == 3 and ppo.default_groups(2048) == 3 and ppo.default_groups(4) == 1


def test_killing_the_parent_takes_the_ranks_with_it(tmp_path):
    """A driver time-out SIGTERMs (or SIGKILLs) `python bench.py --gpus N`: the launcher and its ranks must not be left holding the GPUs.
    The child here is a sleeper that writes its pid; the parent runs launch.run_in_own_group on it and is then killed both ways."""
    import signal
    import time
    for sig in (signal.SIGTERM, signal.SIGKILL):
        pidfile = tmp_path / f"pid_{int(sig)}"
        child = f"import os,time; open({str(pidfile)!r},'w').write(str(os.getpid())); time.sleep(120)"
        parent = ("import sys; sys.path.insert(0, %r); from track_mjx_amd import launch; import os; "
                  "sys.exit(launch.run_in_own_group([sys.executable, '-c', %r], dict(os.environ)))") % (str(ROOT), child)
        p = subprocess.Popen([sys.executable, "-c", parent])
        t0 = time.time()
        while not pidfile.exists() or not pidfile.read_text():
            assert time.time() - t0 < 30 and p.poll() is None
            time.sleep(0.05)
        cpid = int(pidfile.read_text())
        os.kill(p.pid, sig)
        p.wait(timeout=30)
        t0 = time.time()
        while True:
            try:
                os.kill(cpid, 0)
                st = Path(f"/proc/{cpid}/stat").read_text().split()[2]
                if st == "Z":
                    break
            except (ProcessLookupError, FileNotFoundError):
                break
            assert time.time() - t0 < 15, f"child {cpid} survived its parent's {sig!r}"
            time.sleep(0.05)
