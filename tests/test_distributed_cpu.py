"""CPU, world_size 2 over gloo: env sharding, the single-bucket gradient all-reduce (C1) and the normaliser's
cross-rank statistics (C2) give exactly what one process computes on the concatenated data."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from track_mjx_amd.agent.networks import RunningStatistics, ValueNet
from track_mjx_amd.agent.ppo import FlatGrads, shard_range


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = ValueNet(12, (16, 16))
    fg = FlatGrads(list(net.parameters()))
    g = torch.Generator().manual_seed(123)
    x = torch.randn((8, 12), generator=g)
    lo, hi = shard_range(8, rank, world)
    fg.zero()
    (net(x[lo:hi]) ** 2).mean().backward()
    fg.all_reduce_mean()
    norm = fg.clip_by_global_norm(1e-3)
    rs = RunningStatistics(12, "cpu")
    rs.update(x[lo:hi], group=dist.group.WORLD)
    rs.update(x[lo:hi] * 2 + 1, group=dist.group.WORLD)
    if rank == 0:
        q.put((fg.flat.clone().numpy(), float(norm), rs.mean.numpy(), rs.std.numpy(), float(rs.count)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, norm, mean, std, count = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = ValueNet(12, (16, 16))
    fg = FlatGrads(list(net.parameters()))
    x = torch.randn((8, 12), generator=torch.Generator().manual_seed(123))
    fg.zero()
    # mean of per-shard means == mean over the whole batch (equal shard sizes)
    (0.5 * ((net(x[:4]) ** 2).mean() + (net(x[4:]) ** 2).mean())).backward()
    ref_norm = float(torch.linalg.vector_norm(fg.flat))
    fg.clip_by_global_norm(1e-3)
    np.testing.assert_allclose(flat, fg.flat.numpy(), rtol=1e-5, atol=1e-9)
    assert abs(norm - ref_norm) < 1e-5 * max(1, ref_norm)
    assert abs(np.linalg.norm(flat) - 1e-3) < 1e-8          # clip_by_global_norm: scaled down to max_norm
    rs = RunningStatistics(12, "cpu")
    rs.update(x); rs.update(x * 2 + 1)
    np.testing.assert_allclose(mean, rs.mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(std, rs.std.numpy(), rtol=1e-5, atol=1e-6)
    assert count == 16.0


def test_shard_range_partitions_envs():
    assert [shard_range(32768, r, 8) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    try:
        shard_range(10, 0, 3)
        assert False
    except ValueError:
        pass
