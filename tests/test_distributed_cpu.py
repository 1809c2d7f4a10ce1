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
    rs.update(x[lo:hi])                      # default process group, as PPOLearner.update calls it (group=None)
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


# ---- a full PPOLearner.update over 2 gloo ranks: normaliser (C2), per-minibatch loss -> gradients -> all-reduce-mean (C1) ->
# clip_by_global_norm -> Adam, against one process that computes every shard's gradients itself and averages them
_T, _NLOC, _OBS, _REF, _NU = 3, 4, 24, 16, 3
_NETS = dict(encoder_layers=(12,), decoder_layers=(10,), critic_layers=(8,), latents=4)


def _shard_data(rank):
    g = torch.Generator().manual_seed(77 + rank)
    rows = 2 * _NLOC
    return {"observation": torch.randn((_T, rows, _OBS), generator=g) * 2 + 0.5, "raw_action": torch.randn((_T, rows, _NU), generator=g),
            "log_prob": torch.randn((_T, rows), generator=g) * 0.1 - 2, "reward": torch.randn((_T, rows), generator=g),
            "discount": (torch.rand((_T, rows), generator=g) > 0.1).float(), "truncation": (torch.rand((_T, rows), generator=g) > 0.9).float(),
            "next_observation_last": torch.randn((rows, _OBS), generator=g)}


def _perm(rank):
    return lambda upd, rows: torch.randperm(rows, generator=torch.Generator().manual_seed(1000 + 10 * rank + upd))


def _make_learner(world_batch, groups=1):
    from tests.common import StubEnv, torch_gae
    from track_mjx_amd.agent.ppo import PPOLearner
    # `groups` env groups per rank (rollout_groups of train.py / bench.py --pipeline): the rank's envs as a LIST of equal parts
    # (three groups: the bench / train default — of UNEQUAL size when the count does not divide: here 2 + 1 + 1 envs)
    sizes = [_NLOC // groups] * groups if _NLOC % groups == 0 else [_NLOC - (groups - 1)] + [1] * (groups - 1)
    assert sum(sizes) == _NLOC and len(sizes) == groups
    envs = StubEnv(_NLOC, _OBS, _REF, _NU) if groups == 1 else [StubEnv(k, _OBS, _REF, _NU) for k in sizes]
    ln = PPOLearner(envs, **_NETS, unroll_length=_T, batch_size=world_batch, num_minibatches=2, num_updates_per_batch=2,
                    learning_rate=1e-2, use_graph=False, seed=3)
    ln.gae_fn = torch_gae
    return ln


def _seeded(ln, rank):
    """The loss draws fresh noise (entropy sample, latent eps) from torch's global generator: seed it per (rank, call) so that the
    worker processes and the single reference process see the same draws."""
    orig, calls = ln._minibatch_grads, [0]

    def f(idx, kl_w):
        torch.manual_seed(9000 + 100 * rank + calls[0])
        calls[0] += 1
        return orig(idx, kl_w)
    ln._minibatch_grads = f


def _learner_worker(rank, world, port, q, groups=1, single_bucket=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["TMJX_BUCKET_OVERLAP"] = "0" if single_bucket else "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ln = _make_learner(4 * world, groups)         # global batch_size; every rank takes batch_size / world rows of each minibatch
    assert ln.world == world and ln.local_batch == 4 and ln.unrolls == 2 and ln.n_local == _NLOC and len(ln.envs) == groups
    # C1 runs as TWO buckets (value network | policy) unless switched off: the split sits where the value network's parameters begin
    assert ln.overlap_c1 == (not single_bucket) and 0 < ln._bucket_split < ln.grads.flat.numel()
    assert ln._bucket_split == sum((r * pc + 3) // 4 * 4 for _, r, _, pc in ln.grads.segs[:ln._n_policy_params])
    for k, v in _shard_data(rank).items():
        ln.buf[k].copy_(v)
    ln.perm_fn = _perm(rank)
    _seeded(ln, rank)
    ln.update()
    n = ln.normalizer
    q.put((rank, ln.opt.flat.clone().numpy(), n.mean.numpy().copy(), n.std.numpy().copy(), float(n.count)))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world,groups", [(2, 1), (2, 2), (2, 3), (8, 1), (8, 3)])
def test_learner_update_over_gloo_ranks_matches_manual_gradient_average(world, groups):
    """groups = 2 / 3: `world` ranks x two / three env groups per rank (the layout of the 8-GPU bench: three groups on every rank, of unequal size when the
    env count does not divide) — the roll-out buffer rows
    of a rank are its groups' slices side by side, the update must not depend on how the rank's envs are grouped.
    world = 8: BASELINE configs[2]'s rank count (32 768 envs on 8 GPUs; reference: one pmap over 8 local devices, ppo.py:409,477-480,621-623) —
    every rank takes batch_size / 8 rows of each minibatch, C1 averages eight gradients, C2 sums eight shards' statistics."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 37 * groups + 101 * world
    procs = [ctx.Process(target=_learner_worker, args=(r, world, port, q, groups)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r = q.get(timeout=300)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # replicas stay identical (the reference asserts this: ppo.py:805 pmap.assert_is_replicated)
    for r in range(1, world):
        assert np.array_equal(got[0][0], got[r][0]) and np.array_equal(got[0][1], got[r][1]) and got[0][3] == got[r][3] == world * _T * 2 * _NLOC
    # one process: a learner per shard (same init), normaliser over all shards' observations, gradients averaged by hand
    L = [_make_learner(4) for _ in range(world)]
    data = [_shard_data(r) for r in range(world)]
    for r, (ln, d) in enumerate(zip(L, data)):
        for k, v in d.items():
            ln.buf[k].copy_(v)
        _seeded(ln, r)
    allobs = torch.cat([d["observation"].reshape(-1, _OBS) for d in data], 0)
    for ln in L:
        ln.normalizer.update(allobs, distributed=False)
    np.testing.assert_allclose(got[0][1], L[0].normalizer.mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[0][2], L[0].normalizer.std.numpy(), rtol=1e-5, atol=1e-6)
    rows = 2 * _NLOC
    for upd in range(2):
        perms = [_perm(r)(upd, rows) for r in range(world)]
        for mb in range(2):
            flats = []
            for r, ln in enumerate(L):
                ln._minibatch_grads(perms[r][mb * 4:(mb + 1) * 4], ln.kl_weight)
                flats.append(ln.grads.flat.clone())
            mean = torch.stack(flats, 0).sum(0) / world
            for ln in L:
                ln.grads.flat.copy_(mean)
                ln.opt.step()
    ref = L[0].opt.flat.numpy()
    np.testing.assert_allclose(got[0][0], ref, rtol=2e-4, atol=2e-6)
    moved = np.abs(ref - _make_learner(4).opt.flat.numpy()).max()
    assert moved > 1e-3, "the update must have moved the parameters"


def test_bucketed_gradient_all_reduce_is_bit_identical_to_the_single_bucket():
    """C1 as two all-reduces (the value network's bucket first: on the GPU it is issued behind the value network's backward graph and travels
    while the policy's backward pass runs — PPOLearner._capture_split) against ONE all-reduce of the whole flat buffer
    (TMJX_BUCKET_OVERLAP=0): an element-wise mean either way, so two gloo ranks must end on the same bits.
    Reference: track_mjx/agent/mlp_ppo/ppo.py:621-623 (the pmean inside the jitted step, where XLA overlaps it)."""
    ctx = mp.get_context("spawn")
    res = {}
    for single in (False, True):
        q = ctx.Queue()
        port = 33500 + (os.getpid() % 2000) + (11 if single else 0)
        procs = [ctx.Process(target=_learner_worker, args=(r, 2, port, q, 1, single)) for r in range(2)]
        for p in procs:
            p.start()
        got = dict()
        for _ in range(2):
            r = q.get(timeout=180)
            got[r[0]] = r[1:]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert np.array_equal(got[0][0], got[1][0])
        res[single] = got[0][0]
    assert np.array_equal(res[False], res[True])
