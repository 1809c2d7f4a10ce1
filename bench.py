#!/usr/bin/env python3
"""bench.py — env-steps/sec of the rodent tracking hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): rodent tracking, 4096 envs per GPU, PPO with 2x256 encoder/decoder/critic MLPs,
fp32, synthetic reference clips.  One "step" = one PPO training step as the reference defines it
(track_mjx/agent/mlp_ppo/ppo.py:320-395): batch_size*num_minibatches/num_envs = 4 unrolls of unroll_length = 20
env steps on every env (each = 10 physics substeps + reward/obs), the normaliser update, and
num_updates_per_batch*num_minibatches = 64 minibatch SGD updates with the gradient all-reduce.
value = env steps collected by all ranks / wall time (the reference's own `training/sps`, ppo.py:427-431).
Weak scaling: 4096 envs and 1024 minibatch rows per GPU; global batch_size = 1024*N.
`--config cfg3` = BASELINE configs[2] as SURVEY.md §8 d2 defines it (32 768 envs on 8 GPUs, the reference's batch_size 2048): 256 minibatch rows per GPU
(5 120-row GEMMs), ONE unroll + 64 minibatch steps per training step, gradient all-reduce after each; on one GPU it runs one rank's share with the
collectives issued on a one-rank RCCL group; `--gpus N --config cfg3` uses batch_size = 256*N.

Usage: python bench.py [--gpus N --steps K --warmup W]
N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`: RANK / WORLD_SIZE are then in the environment) or bench.py does it itself: `python bench.py --gpus N` starts that
command as a CHILD process before anything touches the GPU and relays rank 0's JSON line (track_mjx_amd/launch.py; the reference
reaches all local devices from one process through jax.pmap, track_mjx/agent/mlp_ppo/ppo.py:409).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# dmabuf IPC: RCCL across processes needs it on this driver (the launchers export it; a rank started some other way must have it before HIP initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ALGO_BYTES_PER_ENV_STEP = 15644        # SURVEY.md §8 d4 / BASELINE.md §2: 3911 four-byte words per env step (K2+K3 together)
# of which the physics kernel (K2) itself moves: reads qpos..time 259 + action 38, writes qpos..time 259 + xpos 204 +
# torso xmat 9 + qfrc_actuator 73 = 842 words (DESIGN.md "Kernels")
K2_ALGO_BYTES_PER_ENV_STEP = 842 * 4
HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md)
ENVS_PER_GPU = 4096
FULL_NETS = dict(encoder_layer_sizes=[1024, 512, 512, 512, 512], decoder_layer_sizes=[512, 512, 512, 256, 256],
                 critic_layer_sizes=[512, 512, 512, 512, 512, 256])            # rodent-full-clips.yaml:50-57
# BASELINE.json configs: cfg2 = configs[1] is the line the metric is quoted on
SMALL_NETS = dict(encoder_layer_sizes=[256, 256], decoder_layer_sizes=[256, 256], critic_layer_sizes=[256, 256])
# `rows_per_gpu`: minibatch rows (envs) per GPU and SGD step = the reference's batch_size / local devices.  cfg2 is this file's weak-scaling default
# (1024 rows per GPU at any N).  cfg3 = BASELINE configs[2] as SURVEY.md §8 d2 defines it: 32 768 envs on 8 GPUs with the reference's batch_size = 2048,
# i.e. ONE RANK'S SHARE is 4096 envs, 256 rows per minibatch (5 120-row GEMMs), one unroll and 64 minibatch steps (each with its gradient
# all-reduce) per training step; `--config cfg3` on one GPU runs exactly that share with C1 / C2 issued on a one-rank RCCL group.
CONFIGS = {
    "cfg2": dict(envs_per_gpu=4096, n_clips=64, nets=SMALL_NETS, rows_per_gpu=1024,
                 matmul_dtype=None, random_clips=False, label="2x256 intention policy + critic, fp32"),
    "cfg3": dict(envs_per_gpu=4096, n_clips=64, nets=SMALL_NETS, rows_per_gpu=256, force_collectives=True,
                 matmul_dtype=None, random_clips=False, label="one rank's share of 32768 envs / 8 GPUs at the reference's batch_size 2048 (256 minibatch rows per GPU), 2x256 nets, fp32, RCCL gradient all-reduce every minibatch step"),
    "cfg4": dict(envs_per_gpu=4096, n_clips=64, nets=FULL_NETS, rows_per_gpu=1024, matmul_dtype=None, random_clips=False,
                 label="rodent-mc-intention nets (enc 1024-512x4, dec 512x3-256x2, critic 512x5-256), fp32"),
    "cfg5": dict(envs_per_gpu=8192, n_clips=1024, nets=FULL_NETS, rows_per_gpu=2048, matmul_dtype="bf16", random_clips=True,
                 label="1024-clip table (per-env clip gather), rodent-mc-intention nets as a bf16 MLP on MFMA: GEMM inputs, hidden activations AND the saved pre-activations z in bf16, fp32 accumulate, fp32 master weights / loss head / optimiser"),
}


def cpu_baseline(blob, clip, seconds_budget: float = 15.0, action_scale: float = 0.3, n_envs: int | None = None):
    """The oracle (CPU restatement, kind "port") stepping the same kind of workload on the host cores.  `action_scale`: actions are
    clip(action_scale * N(0, 1), -1, 1) — 0.3 is the roll-out-only leg's regime (the cost of an env-step depends on it: DESIGN.md §7).
    `n_envs`: BASELINE.md §3 asks for N_env = 1 (one thread: a scalar port) and N_env = 4096 (OpenMP over envs on every host core)."""
    import numpy as np
    from oracle.oracle import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota (containers) bounds the usable cores below the affinity mask
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except Exception:
        pass
    O = Oracle(blob, "f32")
    O.set_clips(clip.as_dict())
    n = max(cores * 4, 16) if n_envs is None else int(n_envs)
    threads = min(cores, n)
    envs = O.new_envs(n)
    rng = np.random.default_rng(0)
    for e in range(n):
        O.env_reset(envs, e, e % clip.position.shape[0], e % 44, rng.uniform(-1e-3, 1e-3, 74), rng.uniform(-1e-3, 1e-3, 73))
    acts = np.clip(action_scale * rng.normal(size=(n, 38)), -1, 1)
    if n <= 256:
        O.env_step_batch(envs, n, acts, threads)  # warm up threads
    t0, steps = time.time(), 0
    while steps == 0 or time.time() - t0 < seconds_budget:
        O.env_step_batch(envs, n, acts, threads)
        steps += n
    dt = time.time() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "host_cores": cores, "n_envs": n, "kind": "port", "action_scale": action_scale,
            "sample": f"{n} envs x {steps // n} control steps (10 substeps each), actions clip({action_scale} * N(0,1)), {threads} OpenMP thread(s) over envs, oracle/liboracle_f32.so"}


def so_build_id() -> str:
    """Build id of the HIP library this run loads (hip.build_id: hash of the sources it was compiled from): ties the numbers read from
    profiles/*.json to a build."""
    from track_mjx_amd import hip
    return hip.build_id()


def mjx_cpu_probe() -> str:
    """BASELINE.md §3.2 / SURVEY §8 d6: MJX on jax[cpu] is the baseline north_star names; it is timed only where it imports."""
    try:
        import jax  # noqa: F401
        import mujoco.mjx  # noqa: F401
        return "available (not timed: the reference env cannot travel to the GPU box)"
    except Exception as e:  # noqa: BLE001
        return f"unavailable ({type(e).__name__}: {e})"


_CHILDREN: list = []        # child benches in flight (other_configs): the signal handler ends their process groups before it exits


def other_configs(timeout_s: float = 150.0) -> dict:
    """Short runs of `bench.py --config cfg3 / cfg4 / cfg5` as child processes (5 timed steps each); the fields of their lines that matter.
    Every child runs in a process group of its own so that a signal that ends this process can end it too (killpg in main's handler)."""
    import signal
    import subprocess
    res = {}
    for cfg in ("cfg3", "cfg4", "cfg5"):
        cmd = [sys.executable, str(ROOT / "bench.py"), "--config", cfg, "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-rollout-only"]
        try:
            p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            _CHILDREN.append(p)
            try:
                so, se = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                so, se = p.communicate()
                res[cfg] = {"error": f"timed out after {timeout_s:.0f} s"}
                continue
            finally:
                if p.poll() is not None and p in _CHILDREN:
                    _CHILDREN.remove(p)
            line = [ln for ln in so.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                res[cfg] = {"error": f"rc {p.returncode}: {se[-300:]}"}
                continue
            o = json.loads(line[-1])
            c = o["config"]
            res[cfg] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "dtype": o["dtype"], "workload": c["workload"],
                        "rollout_ms_per_step": c["rollout_ms_per_step"], "sgd_ms_per_minibatch_step": c["sgd_ms_per_minibatch_step"], "steps": o["steps"],
                        "minibatch_rows_x_unroll": c["minibatch_gemm_rows"], "ranks_seen": c["ranks_seen"], "collectives": c["collectives"],
                        "mlp_gemm_inputs": c["mlp_gemm_inputs"], "roofline_mfma_frac": o["roofline_mfma"]["frac"]}
        except Exception as e:  # noqa: BLE001 — a report beside the headline, never a reason to lose it
            res[cfg] = {"error": f"{type(e).__name__}: {e}"}
    return res


def _pmc_pass(counters: list[str], script_args: list[str], timeout_s: float = 90.0):
    """ONE child `rocprofv3 --pmc <counters> --kernel-trace` pass over `python <script_args>` (the interpreter directly behind `--`, a process group of its
    own, killed on time-out): {kernel name up to '(': {counter: [value per dispatch, in dispatch order]}} or None on any failure."""
    import collections
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    tmp = tempfile.mkdtemp(prefix="tmjx_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", f"{tmp}/p", "--", sys.executable, *script_args]
        p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, start_new_session=True)
        _CHILDREN.append(p)
        try:
            p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            return None
        finally:
            if p in _CHILDREN:
                _CHILDREN.remove(p)
        files = glob.glob(f"{tmp}/p/**/*_counter_collection.csv", recursive=True)
        if p.returncode != 0 or not files:
            return None
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(max(files, key=lambda f: os.path.getmtime(f)))):
            d = per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
            d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])     # (one row per XCD / SE instance: summed per dispatch)
        return {k: {c: [v[i] for i in sorted(v)] for c, v in cs.items()} for k, cs in per.items()}
    except Exception:  # noqa: BLE001 — a report beside the headline, never a reason to lose it
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


SQ_K2_COUNTERS = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES"]


def live_pmc_traffic(timeout_s: float = 90.0):
    """HBM traffic of the env.step kernels AND the vector-pipe counters of K2 from the PMC counters, collected BY THIS RUN: two child `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE cannot share a pass — 3 + 2 of the 4 TCC slots: MI355X_MICROARCH.md "rocprofv3 PMC slots"; the four SQ counters ride with the
    FETCH_SIZE pass, the SQ block has 8 slots of its own) over tools/time_step.py (4096 envs in one launch, 4 steps), traffic counters in KB per dispatch,
    FETCH_SIZE doubled — the gfx950 correction that guide prescribes.  Exactly what tools/profile_gpu.sh + tools/pmc_summary.py / tools/sq_counters.sh +
    tools/sq_summary.py stamp into profiles/pmc_traffic.json / sq_counters.json (the fallback when rocprofv3 is missing or a pass fails: the line says
    which it carries).  Returns None on any failure."""
    step_kernels = ("void k_physics_wave", "k_rec_in", "k_rec_out", "k_step_parts", "k_window", "k_obs", "k_post_parts", "k_post", "k_autoreset")
    script = [str(ROOT / "tools" / "time_step.py"), "--steps", "4", "--scale", "0.3"]
    try:
        first = _pmc_pass(["FETCH_SIZE", *SQ_K2_COUNTERS], script, timeout_s)
        sq_rode_along = first is not None
        if first is None:
            first = _pmc_pass(["FETCH_SIZE"], script, timeout_s)          # (should the two blocks' counters not combine on some driver: the traffic pass alone)
        second = _pmc_pass(["WRITE_SIZE"], script, timeout_s) if first is not None else None
        if first is None or second is None:
            return None

        def steady(v):            # steady-state env.step launches (the first ones belong to reset / warm-up)
            return v[3:] if len(v) > 4 else v
        out: dict = {}
        for kind, res, ctr in (("fetch", first, "FETCH_SIZE"), ("write", second, "WRITE_SIZE")):
            for k, cs in res.items():
                if k.startswith(step_kernels) and ctr in cs:
                    vv = steady(cs[ctr])
                    out.setdefault(k.replace("void ", ""), {})[kind] = sum(vv) / len(vv) * 1024.0
        kern = {k: {"fetch_bytes_raw": v.get("fetch", 0.0), "fetch_bytes_corrected": 2 * v.get("fetch", 0.0), "write_bytes": v.get("write", 0.0)} for k, v in out.items()}
        dom = "k_physics_wave<true>"
        if dom not in kern or "fetch" not in out[dom] or "write" not in out[dom]:
            return None
        ret = {"kernels": kern, "envs_per_launch": ENVS_PER_GPU, "hbm_bytes_per_launch": kern[dom]["fetch_bytes_corrected"] + kern[dom]["write_bytes"],
               "hbm_bytes_per_launch_all_step_kernels": sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in kern.values()),
               "so_build_id": so_build_id(), "collected_by_this_run": True, "sq": None}
        k2 = first.get("void k_physics_wave<true>") if sq_rode_along else None
        if k2 and all(c in k2 for c in SQ_K2_COUNTERS):
            # per wave (= env) and substep, as tools/sq_summary.py: pipe busy = SQ_INSTS_VALU x 2 cycles x waves per SIMD / resident cycles (SQ_WAVE_CYCLES ticks every
            # 4 clocks); lane occupancy = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
            c = {n: sum(steady(k2[n])) / len(steady(k2[n])) / (ENVS_PER_GPU * 10) for n in SQ_K2_COUNTERS}
            from track_mjx_amd.agent import ppo as _ppo
            wps = _ppo.RESIDENT_ENVS_PER_CU / 4.0
            ret["sq"] = {"valu_insts_per_wave_substep": c["SQ_INSTS_VALU"], "valu_pipe_busy": 2.0 * c["SQ_INSTS_VALU"] * wps / (4.0 * c["SQ_WAVE_CYCLES"]),
                         "lane_occupancy": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), "waves_per_simd": wps}
        return ret
    except Exception:  # noqa: BLE001 — a report beside the headline, never a reason to lose it
        return None


def live_mfma_util(timeout_s: float = 90.0):
    """Matrix-pipe utilisation inside the learner's GEMM / chain kernels over the SGD half of a cfg2 training step (tools/sgd_step.py, eager launches), collected
    BY THIS RUN with one child `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass: utilisation = MFMA busy cycles / (4 SIMDs x CU busy cycles),
    summed per kernel family (tools/mfma_summary.py's formula; fallback: the stamped profiles/mfma_counters.json)."""
    res = _pmc_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"], [str(ROOT / "tools" / "sgd_step.py"), "--config", "cfg2"], timeout_s)
    if not res:
        return None
    fam: dict = {}
    for k, cs in res.items():
        name = ("chain kernels (k_chain_fwd / k_chain_bwd)" if "k_chain_" in k else "k_gemm_dw (weight gradient)" if "k_gemm_dw" in k else
                "k_gemm_act (forward / input gradient)" if "k_gemm_act" in k else None)
        if name and "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CU_CYCLES" in cs:
            f = fam.setdefault(name, [0.0, 0.0])
            f[0] += sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]); f[1] += sum(cs["SQ_BUSY_CU_CYCLES"])
    if not fam:
        return None
    tot = [sum(f[0] for f in fam.values()), sum(f[1] for f in fam.values())]
    return {"families": {k: round(f[0] / max(4 * f[1], 1.0), 4) for k, f in fam.items()}, "all": round(tot[0] / max(4 * tot[1], 1.0), 4), "so_build_id": so_build_id()}


def parity_summary():
    """What the north_star's "qpos / qvel within 1e-5 rel" means for this build, read from the newest profiles/r*_parity_strict.log (the printed summary of
    tests/test_gpu_parity_strict.py: 64 envs x 40 teacher-forced substeps per action regime against the float64 oracle)."""
    import re
    logs = sorted((ROOT / "profiles").glob("r*_parity_strict.log"))
    out = {"oracle": "unpinned (own restatement of mujoco-mjx 3.3.2: the reference ships no fixtures and jax / mujoco cannot be imported here)",
           "contract": "median over env-substeps <= 1e-5 relative to a float64 run of the oracle, one substep from an identical state; NOT every env-substep "
                       "(a 5-iteration non-converged CG amplifies fp32 rounding: the float32 restatement of MJX's own dense path does no better)"}
    if not logs:
        return out
    txt = logs[-1].read_text()
    fr, fo, med = [], [], []
    for blk in re.split(r"\[scale ", txt)[1:4]:
        m = re.search(r"qvel: HIP against the FLOAT32 oracle.*?float64 oracle: HIP ([0-9.]+), float32 oracle ([0-9.]+)", blk)
        q = re.search(r"qvel: HIP worst env \S+ \(float32 oracle \S+\); 99th pct \S+ \(\S+\); median (\S+) \((\S+)\)", blk)
        if m:
            fr.append(float(m.group(1))); fo.append(float(m.group(2)))
        if q:
            med.append(float(q.group(1)))
    out.update({"action_scales": [0.03, 0.3, 1.0], "qvel_within_1e-5": fr, "qvel_within_1e-5_float32_oracle": fo, "qvel_median_rel_err": med, "source": f"profiles/{logs[-1].name}"})
    return out


def dry_run_ranks(args) -> None:
    """`--dry-run-ranks`: the launcher plumbing without a GPU — every rank joins a gloo group, all-reduces a one, rank 0 prints the
    line's skeleton (tests/test_launch.py drives `bench.py --gpus 2 --dry-run-ranks` through the self-launch branch)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ranks_seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    if int(os.environ.get("RANK", "0")) == 0:
        bc = CONFIGS[args.config]
        n_local = args.envs_per_gpu or bc["envs_per_gpu"]
        batch_size = bc["rows_per_gpu"] * world * n_local // bc["envs_per_gpu"]          # as main() hands it to the learner
        print(json.dumps({"metric": "dry-run-ranks", "n_gpus": world, "asked_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "config": {"ranks_seen": ranks_seen, "parallelism": f"dp{world}", "workload": args.config, "envs_total": n_local * world,
                                     "global_batch": batch_size, "minibatch_rows_per_gpu": batch_size // world,
                                     "unrolls_per_training_step": batch_size * 16 // (n_local * world)}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None, runner=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=0, help="env groups per GPU whose roll-outs are pipelined on separate HIP streams (0 = ppo.default_groups: 3; 1 = off; sizes: ppo.group_sizes)")
    ap.add_argument("--no-rollout-only", action="store_true", help="skip the extra roll-out-only measurement (tools/profile_gpu.sh: keeps the "
                    "rocprofv3 kernel averages those of the timed training steps)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2",
                    help="BASELINE.json configs[1] (default, the headline line) / configs[2] (cfg3: one rank's share per GPU) / configs[3] / configs[4]; the others are extra measurements")
    ap.add_argument("--dry-run-ranks", action="store_true", help="launcher check on CPU: gloo ranks, no GPU work (tests)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the HBM-traffic counters with two child rocprofv3 passes (then the stamped profiles/pmc_traffic.json is what the line carries)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short cfg4 / cfg5 measurements the default N = 1 line carries under config.other_configs")
    args = ap.parse_args(argv)
    from track_mjx_amd import launch
    if launch.needs_spawn(args.gpus):
        # `python bench.py --gpus N` as the driver types it: start the N ranks as a child process (nothing has touched the GPU yet)
        return launch.spawn_ranks(args.gpus, [str(ROOT / "bench.py")], argv, runner=runner)
    if args.dry_run_ranks:
        return dry_run_ranks(args)
    bc = CONFIGS[args.config]
    if args.envs_per_gpu is None:
        args.envs_per_gpu = bc["envs_per_gpu"]
    if os.environ.get("TMJX_PIN_CORES"):        # host-starvation stress (tools/gpu_lab.sh starve): this rank's threads on the named cores, before its first GPU call
        os.sched_setaffinity(0, {int(c) for c in os.environ["TMJX_PIN_CORES"].split(",")})

    import torch
    import torch.distributed as dist
    from track_mjx_amd import config as _config
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.environment import wrap
    from track_mjx_amd.train import build_env

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # TMJX_REHEARSE_ON_ONE_GPU=1: every rank on cuda:0 with gloo collectives on the device tensors — the N > 1 call path of this file (rank
    # seeding, normaliser / gradient all-reduces between graph replays, barriers, max-over-ranks timing, rank-0 reporting) on a ONE-GPU box,
    # where RCCL refuses two ranks on one device.  A rehearsal of the plumbing, never a measurement (the line says so).
    rehearse = bool(os.environ.get("TMJX_REHEARSE_ON_ONE_GPU"))
    device = torch.device("cuda:0" if rehearse else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    # TMJX_COLLECTIVES_ALWAYS=1 under torch.distributed.run with ONE rank: RCCL is initialised and C1 / C2 / the timing reductions are
    # issued on the one-rank group — the multi-GPU call path on a single-GPU box (the numbers equal the plain N=1 run's)
    if bc.get("force_collectives") and world == 1 and not rehearse:
        # cfg3 on ONE GPU is one rank's share of the 8-GPU job: the rank's collectives (C1 after every minibatch step, C2, the timing reductions)
        # are issued on a one-rank RCCL group, started in-process when no launcher provided the rendezvous
        os.environ["TMJX_COLLECTIVES_ALWAYS"] = "1"
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()))
    use_dist = world > 1 or ("RANK" in os.environ and bool(os.environ.get("TMJX_COLLECTIVES_ALWAYS")))
    if use_dist:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    cfg = _config.default_config()
    cfg["network_config"].update(**bc["nets"])
    tc = cfg["train_setup"]["train_config"]
    n_local = args.envs_per_gpu
    # the rank's envs as `--pipeline` equal groups (same clips, same model): the learner pipelines their roll-outs on separate
    # HIP streams so that one group's kernel tail + reward/obs kernels + policy inference run next to the other group's physics
    ngrp = args.pipeline if args.pipeline >= 1 else ppo.default_groups(n_local, device, widest_layer=max(max(v) for k, v in bc["nets"].items() if k.endswith("_layer_sizes")))
    from track_mjx_amd import clips as _clips
    from track_mjx_amd.walker import Rodent
    table = _clips.make_synthetic_clips(Rodent(**cfg["walker_config"]).model, bc["n_clips"], n_frames=cfg["reference_config"]["clip_length"],
                                        mocap_hz=cfg["env_config"]["env_args"]["mocap_hz"])
    sizes = ppo.group_sizes(n_local, ngrp)
    if os.environ.get("TMJX_GROUP_SIZES"):       # experiments: explicit group sizes, e.g. 1408,1408,1280
        sizes = [int(x) for x in os.environ["TMJX_GROUP_SIZES"].split(",")]
        assert sum(sizes) == n_local, (sizes, n_local)
    ngrp = len(sizes)
    envs = [wrap(build_env(cfg, sizes[0], device, reference_clip=table), episode_length=195)]
    envs += [wrap(build_env(cfg, sizes[k], device, reference_clip=table, share_clips_with=envs[0]), episode_length=195) for k in range(1, ngrp)]      # one clip upload per rank
    env = envs[0]
    nc = cfg["network_config"]
    learner = ppo.PPOLearner(envs if ngrp > 1 else env, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"],
                             critic_layers=nc["critic_layer_sizes"], latents=nc["intention_size"], learning_rate=tc["learning_rate"],
                             entropy_cost=tc["entropy_cost"], discounting=tc["discounting"], unroll_length=tc["unroll_length"],
                             batch_size=bc["rows_per_gpu"] * world * n_local // bc["envs_per_gpu"], num_minibatches=tc["num_minibatches"],
                             num_updates_per_batch=tc["num_updates_per_batch"], normalize_observations=True, kl_weight=nc["kl_weight"],
                             seed=0, matmul_dtype=torch.bfloat16 if bc["matmul_dtype"] == "bf16" else None)
    # deterministic synthetic reset inputs: clip = env % 64, start_frame = env % 44; cfg5: clip ~ U{0..1023}
    g = torch.Generator().manual_seed(1 + rank)
    idx = torch.arange(n_local, dtype=torch.int32) + rank * n_local
    starts = [sum(sizes[:k]) for k in range(ngrp + 1)]
    for k, e in enumerate(envs):
        sub = idx[starts[k]:starts[k + 1]]
        per = sizes[k]
        clip_idx = torch.randint(0, bc["n_clips"], (per,), generator=g, dtype=torch.int32) if bc["random_clips"] else (sub % bc["n_clips"]).to(torch.int32)
        learner.states[k] = e.reset(g, clip_idx, start_frame=(sub % 44).to(torch.int32))

    # HIP-event timing of the dominant kernel (k_physics_wave: the 10 physics substeps of one control step for all envs) on
    # the launch stream: env.step issues K2 and K3 as two ABI calls and records events around K2 (environment/task.py)
    for _ in range(args.warmup):
        learner.training_step(1)
    for e in envs:
        e._physics_events = []

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    ranks_seen = 1
    if use_dist:      # one RCCL all-reduce before the timed region: every rank contributes 1 (and the communicator is warm)
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    sync()
    # roll-out / SGD split of the timed steps: HIP events on the learner's main stream around collect() and update()
    split_ev = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev3[0].record(); learner.collect(); ev3[1].record(); learner.update(1); ev3[2].record()
        split_ev.append(ev3)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rollout_ms = sum(a.elapsed_time(b) for a, b, _ in split_ev) / args.steps
    sgd_ms = sum(b.elapsed_time(c) for _, b, c in split_ev) / args.steps
    env_steps = learner.env_steps_per_training_step * args.steps
    ev = [p for e in envs for p in e._physics_events]
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)      # per launch (n_local / ngrp envs each), launches of the groups overlap
    per_launch = n_local / ngrp          # (groups differ by at most 4 envs: ppo.group_sizes)
    for e in envs:
        e._physics_events = None
    # d1 of the measurement contract also asks for the ROLL-OUT-ONLY rate (random actions, no policy, no learner): the same env
    # groups stepped on their own streams, outside the timed region of the training metric; and the physics kernel ALONE on an
    # otherwise idle GPU (one group's launches back to back): the isolated per-launch duration next to the shared-GPU one above
    rollout_only = rollout_err = isolated_ms = None
    if not args.no_rollout_only:
        try:
            gen = torch.Generator(device=device).manual_seed(5)
            acts = [torch.randn((38, e.num_envs), generator=gen, device=device).clamp(-1, 1) * 0.3 for e in envs]
            # the learner's own group streams: new streams could land on a hardware queue that two groups then share (they would serialise)
            streams = list(learner._streams) if getattr(learner, "_streams", None) else [torch.cuda.Stream(device=device) for _ in envs]
            sts = list(learner.states)
            nroll = 40
            for _rep in range(2):                 # first repetition = warm-up
                torch.cuda.synchronize(device)
                tr0 = time.perf_counter()
                for _ in range(nroll):
                    for k, e in enumerate(envs):
                        with torch.cuda.stream(streams[k]):
                            sts[k] = e.step(sts[k], acts[k])
                torch.cuda.synchronize(device)
                rollout_only = n_local * nroll / (time.perf_counter() - tr0)
            e0 = envs[0]
            e0._physics_events = []
            for _ in range(20):
                sts[0] = e0.step(sts[0], acts[0])
            torch.cuda.synchronize(device)
            iso = [a.elapsed_time(b) for a, b in e0._physics_events[4:]]
            isolated_ms = sum(iso) / len(iso)
            e0._physics_events = None
        except Exception as ex:  # noqa: BLE001 — reported in the JSON line and on stderr, never silently dropped
            import traceback
            traceback.print_exc()
            rollout_err = f"{type(ex).__name__}: {ex}"

    if rank == 0:
        def prof(name):
            f = ROOT / "profiles" / name
            try:
                return json.loads(f.read_text()) if f.exists() else None
            except Exception:
                return None
        achieved = ALGO_BYTES_PER_ENV_STEP * per_launch / (kernel_ms * 1e-3) / 1e9
        k2_achieved = K2_ALGO_BYTES_PER_ENV_STEP * per_launch / (kernel_ms * 1e-3) / 1e9
        pmc, sq, flc, mf = prof("pmc_traffic.json"), prof("sq_counters.json"), prof("oracle_flop_count.json"), prof("mfma_counters.json")
        live = None           # (the counters collected by THIS run replace the stamped file's further down, once the headline is safe from a time-out)
        # counter bytes per launch, scaled to this launch's env count: K2 alone, and all kernels of one env.step (K2 + record transposes + K3)
        pscale = per_launch / pmc.get("envs_per_launch", ENVS_PER_GPU) if pmc else None
        k2_traffic = pmc.get("hbm_bytes_per_launch") * pscale if pmc else None
        step_traffic = pmc.get("hbm_bytes_per_launch_all_step_kernels") * pscale if pmc and pmc.get("hbm_bytes_per_launch_all_step_kernels") else None
        rate_rollout = rollout_only if rollout_only else env_steps / elapsed / world
        flops_env_step = flc["flops_per_env_step"] if flc else None
        # MLP GEMM flops of one minibatch step: forward + d input + d weight = 3 x 2 x rows x weights (policy and value), plus the
        # bootstrap value forward
        Pw = sum(p.numel() for p in learner.policy.parameters() if p.dim() == 2)
        Vw = sum(p.numel() for p in learner.value.parameters() if p.dim() == 2)
        rows_mb = learner.local_batch * learner.T
        sgd_steps = learner.num_updates * learner.num_minibatches
        gemm_flops_step = 6.0 * (Pw + Vw) * rows_mb + 2.0 * Vw * learner.local_batch
        mfma_peak = 2500.0 if bc["matmul_dtype"] == "bf16" else 157.3
        mfma_ach = gemm_flops_step * sgd_steps / (sgd_ms * 1e-3) / 1e12
        out = {
            "metric": "env-steps/sec (whole node), rodent task @ 4096 envs/GPU", "value": env_steps / elapsed,
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # (bf16 mode goes beyond SURVEY a16's "bf16 only for GEMM inputs": the pre-activations the backward epilogues re-read are saved as bf16 too —
            # BASELINE configs[4] says "bf16 MLP on MFMA"; the fp32-z form is a build switch of the same sources, -DTMJX_BF16_Z_F32: 4.01 against 3.79 ms per minibatch step, profiles/r05_cfg5_z_variants.txt)
            "dtype": "f32" if bc["matmul_dtype"] is None else f"f32 (physics, reward, loss head, optimiser, master weights) + {bc['matmul_dtype']} MLP: GEMM inputs, hidden activations and saved pre-activations in {bc['matmul_dtype']}, f32 accumulate",
            "data": "synthetic", "so_build_id": so_build_id(), "mjx_cpu": mjx_cpu_probe(),
            "config": {"workload": f"rodent tracking PPO training step ({args.config}): {n_local} envs/GPU, {learner.env_steps_per_training_step // (n_local * learner.T * world)}x{learner.T}-step unrolls (10 physics substeps each) + {sgd_steps} minibatch updates, {bc['label']}",
                       "n_clips": bc["n_clips"], "mlp_gemm_inputs": bc["matmul_dtype"] or "f32",
                       "envs_per_gpu": n_local, "global_batch": learner.local_batch * world, "unroll_length": learner.T,
                       "minibatch_gemm_rows": learner.local_batch * learner.T, "collectives": bool(learner.collectives),
                       "weak_scaling_rule": f"{bc['rows_per_gpu']} minibatch rows and {bc['envs_per_gpu']} envs per GPU at any N (global batch_size = {bc['rows_per_gpu']} x N)",
                       "env_steps_per_step": learner.env_steps_per_training_step, "parallelism": f"dp{world}", "ranks_seen": ranks_seen,
                       **({"rehearsal": "all ranks on cuda:0 with gloo collectives (TMJX_REHEARSE_ON_ONE_GPU): plumbing check, NOT a measurement"} if rehearse else {}),
                       "policy_params": learner.n_params(), "envs_per_physics_launch": per_launch, "concurrent_physics_launches": ngrp,
                       "rollout_ms_per_step": rollout_ms, "sgd_ms_per_step": sgd_ms, "sgd_ms_per_minibatch_step": sgd_ms / sgd_steps,
                       "rollout_only_env_steps_per_s_per_gpu": rollout_only, "rollout_only_error": rollout_err,
                       "note": "one box of the pool differs from the next by about 1-3 % on this line (round 4's final build: 1.446-1.471 M env-steps/s over five runs)"},
            # 15 644 algorithmic bytes per env-step (K2 + K3 together) x the envs of one physics launch / that
            # launch's average duration (HIP events on its launch stream; the env groups' launches share the GPU)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # like with like: `achieved` prices K2 + K3's 15 644 algorithmic bytes, so `traffic` is the counter bytes of ALL kernels of
                         # one env.step (K2, the two record transposes, K3's three launches); K2's own pair sits under "k2_only"
                         "traffic": step_traffic,
                         "traffic_over_algorithmic": (step_traffic / (ALGO_BYTES_PER_ENV_STEP * per_launch)) if step_traffic else None,
                         "traffic_build_id": pmc.get("so_build_id") if pmc else None,
                         "traffic_build_is_this_runs": bool(pmc and pmc.get("so_build_id") == so_build_id()),
                         "traffic_collected_by_this_run": False,
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh (FETCH_SIZE doubled, the gfx950 correction), NOT collected by this run (replaced below when this run's own rocprofv3 passes succeed); bytes per launch scaled to this launch's env count",
                         "kernel": "k_physics_wave (10 physics substeps of one control step, one workgroup per env)",
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * per_launch,
                         "avg_launch_ms": kernel_ms, "avg_launch_ms_is": f"shared-GPU duration: {ngrp} launches of {' / '.join(str(x) for x in sizes)} envs run concurrently (plus the other group's K3 / inference); the HIP events also span the two record-transpose launches (< 1 %)",
                         "concurrent_launches": ngrp,
                         "frac_of_the_concurrent_launches_together": achieved * ngrp / HBM_PEAK_GBS,     # the launches share the GPU: per launch the figure above FALLS when a third group is added although the chip does more
                         "avg_launch_ms_isolated": isolated_ms,
                         "frac_isolated": (ALGO_BYTES_PER_ENV_STEP * sizes[0] / (isolated_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if isolated_ms else None,
                         "k2_only": {"algorithmic_bytes_per_env_step": K2_ALGO_BYTES_PER_ENV_STEP, "algorithmic_bytes_per_launch": K2_ALGO_BYTES_PER_ENV_STEP * per_launch,
                                     "achieved": k2_achieved, "frac": k2_achieved / HBM_PEAK_GBS, "traffic": k2_traffic,
                                     "traffic_over_algorithmic": (k2_traffic / (K2_ALGO_BYTES_PER_ENV_STEP * per_launch)) if k2_traffic else None},
                         "chip_level_frac_at_rollout_only_rate": ALGO_BYTES_PER_ENV_STEP * rate_rollout / 1e9 / HBM_PEAK_GBS,
                         "chip_level_frac_at_training_rate": ALGO_BYTES_PER_ENV_STEP * (env_steps / elapsed / world) / 1e9 / HBM_PEAK_GBS},
            # what actually binds K2: the vector ALU / latency (about 660 flop per algorithmic byte).  Numerator: the instrumented operation
            # count of the reference's (dense MJX) algorithm, frozen in profiles/oracle_flop_count.json; pipe occupancy and lane use from
            # the SQ counters of the same kernel (profiles/sq_counters.json: SQ_INSTS_VALU x 2 cycles / SIMD-cycles)
            "roofline_valu": {"bound": "valu", "unit": "TFLOP/s", "peak": 157.3,
                              "achieved": (flops_env_step * rate_rollout / 1e12) if flops_env_step else None,
                              "frac": (flops_env_step * rate_rollout / 1e12 / 157.3) if flops_env_step else None,
                              "at": "roll-out-only rate" if rollout_only else "training rate",
                              "flops_per_env_step": flops_env_step,
                              "flops_source": "profiles/oracle_flop_count.json: flop-counting build of the CPU oracle (dense MJX formulation: 1.03 Mflop per substep); the kernel's tree-sparse / matrix-free formulation executes fewer",
                              "valu_pipe_busy": sq.get("valu_pipe_busy") if sq else None, "lane_occupancy": sq.get("lane_occupancy") if sq else None,
                              "valu_insts_per_wave_substep": sq.get("SQ_INSTS_VALU_per_wave_substep") if sq else None,
                              "counters_source": "profiles/sq_counters.json (rocprofv3 --pmc SQ_* passes, tools/sq_counters.sh), not collected by this run",
                              "counters_build_id": sq.get("so_build_id") if sq else None,
                              "counters_build_is_this_runs": bool(sq and sq.get("so_build_id") == so_build_id()), "counters_collected_by_this_run": False},
            "roofline_mfma": {"bound": "mfma", "unit": "TFLOP/s", "peak": mfma_peak, "achieved": mfma_ach, "frac": mfma_ach / mfma_peak,
                              "what": f"{sgd_steps} minibatch SGD steps of {rows_mb} rows: GEMM flops (forward + d input + d weight, policy + value nets) / the whole SGD half's time incl. gathers, epilogues, loss head and optimiser (HIP events around update())",
                              "gemm_flops_per_minibatch_step": gemm_flops_step, "sgd_ms_per_minibatch_step": sgd_ms / sgd_steps,
                              "mfma_util_inside_gemm_kernels": mf.get("mfma_util") if mf else None, "mfma_util_collected_by_this_run": False,
                              "mfma_util_source": "profiles/mfma_counters.json (tools/mfma_counters.sh), not collected by this run"},
            "parity": parity_summary(),
        }
        # the finished measurement is on disk before any extra runs: a hard kill (SIGKILL, a driver time-out) during the minute of extras below cannot lose it
        # (stdout still carries exactly ONE line, the enriched one, printed by emit())
        for side in (ROOT / "gpurun_out", Path("/tmp")):
            try:
                side.mkdir(exist_ok=True)
                (side / "bench_headline.json").write_text(json.dumps(out) + "\n")
                break
            except OSError:
                continue
        # the extras below (CPU baselines, child runs of cfg4 / cfg5) take a minute: if the caller's time-out ends this process during them
        # (SIGTERM / SIGINT), the headline measured above is printed as it stands instead of being lost — still exactly one JSON line
        import signal
        printed = []

        def emit(*_sig):
            # (SIGTERM / SIGINT are blocked while the line is written: a signal that lands inside print() must not skip the flush)
            if _sig:
                for p in list(_CHILDREN):            # a child bench still on the GPU: end its whole process group first
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except Exception:  # noqa: BLE001
                        pass
            old = signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM, signal.SIGINT})
            try:
                if not printed:
                    if _sig:
                        out["config"]["extras_cut_short_by_signal"] = int(_sig[0])
                    print(json.dumps(out), flush=True)
                    printed.append(1)
            finally:
                if _sig:
                    os._exit(128 + int(_sig[0]))     # cut short: the headline line is out, the exit code says the run did not finish
                signal.pthread_sigmask(signal.SIG_SETMASK, old)
        for _s in (signal.SIGTERM, signal.SIGINT):
            signal.signal(_s, emit)
        # the HBM-traffic counters collected by THIS run (two child rocprofv3 --pmc passes, ~40 s; only the default N = 1 line): the stamped
        # profiles/pmc_traffic.json stays in the line if rocprofv3 is missing or a pass fails
        if world == 1 and args.config == "cfg2" and not args.no_live_pmc and not args.no_cpu_baseline:
            live = live_pmc_traffic()
            if live and live.get("sq"):      # the vector-pipe counters of K2 rode along with the FETCH_SIZE pass: they replace the stamped file's
                out["roofline_valu"].update({"valu_pipe_busy": live["sq"]["valu_pipe_busy"], "lane_occupancy": live["sq"]["lane_occupancy"],
                                             "valu_insts_per_wave_substep": live["sq"]["valu_insts_per_wave_substep"], "counters_build_id": live["so_build_id"],
                                             "counters_build_is_this_runs": True, "counters_collected_by_this_run": True,
                                             "counters_source": "child `rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES` pass of THIS run over tools/time_step.py (4096 envs in one launch, 0.3-scaled actions)"})
            mfl = live_mfma_util()
            if mfl:
                out["roofline_mfma"].update({"mfma_util_inside_gemm_kernels": mfl["all"], "mfma_util_by_kernel_family": mfl["families"], "mfma_util_collected_by_this_run": True,
                                             "mfma_util_source": "child `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass of THIS run over tools/sgd_step.py --config cfg2"})
            if live:
                sc = per_launch / live["envs_per_launch"]
                r, k2t, stt = out["roofline"], live["hbm_bytes_per_launch"] * sc, live["hbm_bytes_per_launch_all_step_kernels"] * sc
                r.update({"traffic": stt, "traffic_over_algorithmic": stt / (ALGO_BYTES_PER_ENV_STEP * per_launch), "traffic_build_id": live["so_build_id"],
                          "traffic_build_is_this_runs": True, "traffic_collected_by_this_run": True,
                          "traffic_source": "two child `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of THIS run over tools/time_step.py (4096 envs in one launch; KB per dispatch, FETCH_SIZE doubled: the gfx950 correction), bytes per launch scaled to this launch's env count",
                          "traffic_kernels": {k: v["fetch_bytes_corrected"] + v["write_bytes"] for k, v in live["kernels"].items()}})
                r["k2_only"].update({"traffic": k2t, "traffic_over_algorithmic": k2t / (K2_ALGO_BYTES_PER_ENV_STEP * per_launch)})
        if world == 1 and not args.no_cpu_baseline:
            try:
                # same action regime as the roll-out-only leg (0.3 * N(0,1)); the full-scale regime of BASELINE config 1 beside it
                # N_env = 4096 on every host core (the headline `value`) and N_env = 1 on one thread
                out["cpu_baseline"] = cpu_baseline(env._blob, env._reference_clips, 10.0, 0.3, n_envs=ENVS_PER_GPU)
                out["cpu_baseline"]["n_env_1"] = cpu_baseline(env._blob, env._reference_clips, 4.0, 0.3, n_envs=1)
                out["cpu_baseline"]["full_scale_actions"] = cpu_baseline(env._blob, env._reference_clips, 4.0, 1.0)
                out["cpu_baseline"]["mjx_cpu"] = out["mjx_cpu"]
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        # BASELINE configs[2] (one rank's share) / configs[3] / configs[4] beside the headline line (driver-visible: the default run is the only one the driver makes): each as a
        # CHILD process of this script after the timed region (own process: its failure or time-out cannot touch the headline), 5 timed steps
        if world == 1 and args.config == "cfg2" and not args.no_other_configs and not args.no_cpu_baseline:
            out["config"]["other_configs"] = other_configs()
        emit()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
