#!/usr/bin/env python3
"""N ranks of the REAL training step (HIP physics, hipGraph inference and SGD step, fused loss head) on ONE GPU with gloo collectives on the device
tensors: after a few steps every rank must hold bit-identical parameters, optimiser moments and normaliser statistics although each rank rolled
out different envs.  RCCL refuses two ranks on one device, so this is as close to `--gpus N` as a one-GPU box gets (RCCL itself: tests/test_gpu_rccl.py,
one rank).  Launch (GPU box): python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/two_rank_sync_check.py [steps]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.common import make_env_and_oracle  # noqa: E402
from track_mjx_amd.agent import ppo  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
envs = [make_env_and_oracle(num_envs=n, n_clips=4, wrappers=True, seed=100 * rank + k)[0] for k, n in enumerate((48, 44, 36))]   # three unequal env groups
L = ppo.PPOLearner(envs, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5, batch_size=32 * world,      # (global, like the reference's: batch_size * num_minibatches % num_envs == 0)
                   num_minibatches=4, num_updates_per_batch=2, seed=3, normalize_observations=True)
assert L.collectives and L.world == world and L.rank == rank, (L.collectives, L.world, L.rank)
for k, e in enumerate(envs):
    L.states[k] = e.reset(torch.Generator().manual_seed(1000 * rank + k))
first = L.opt.flat.clone()
for it in range(steps):
    m = L.training_step(it)
torch.cuda.synchronize()
assert all(bool(torch.isfinite(v).all()) for v in m.values()), m


def gathered(t):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t.contiguous())
    return out


report = {}
for name, t in (("parameters", L.opt.flat), ("adam exp_avg", L.opt.exp_avg), ("adam exp_avg_sq", L.opt.exp_avg_sq), ("normaliser mean", L.normalizer.mean),
                ("normaliser std", L.normalizer.std), ("normaliser count", L.normalizer.count.reshape(1).float())):
    g = gathered(t)
    report[name] = all(torch.equal(g[0], x) for x in g[1:])
obs = gathered(L.buf["observation"][0, :16].contiguous())                # the ranks did NOT see the same data
different_data = not any(torch.equal(obs[0], x) for x in obs[1:])
moved = float((L.opt.flat - first).abs().max())
if rank == 0:
    print(f"{world} ranks on one GPU, {steps} training steps, graph={L._graph is not None}: " + ", ".join(f"{k} {'identical' if v else 'DIFFER'}" for k, v in report.items())
          + f"; ranks rolled out different observations: {different_data}; max |parameter change| {moved:.3e}", flush=True)
ok = all(report.values()) and different_data and moved > 0
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
