"""RCCL on the hardware that is there: a ONE-rank nccl (= RCCL) process group with the collectives of the training step forced on
(TMJX_COLLECTIVES_ALWAYS=1).  All-reduces over one rank are identities, so the learner must end up where the plain single-process learner
does — what this covers is the call path no CPU test can: RCCL initialisation on the GPU, C1 (the flat-gradient all-reduce) issued between
replays of the captured SGD-step hipGraph with the RCCL watchdog thread alive (capture_error_mode="thread_local"), and C2 (the running
statistics' sums) between the two K6 kernels.  Reference: track_mjx/agent/mlp_ppo/ppo.py:357-361 (pmean of the statistics), :409 (pmean of
the gradients).  The N > 1 arithmetic is covered on CPU by tests/test_distributed_cpu.py (gloo, world_size 2)."""
import os
import socket

import pytest
import torch

from tests.common import make_env_and_oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _learner(seed_env):
    from track_mjx_amd.agent import ppo
    envs = [make_env_and_oracle(num_envs=64, n_clips=4, wrappers=True, seed=seed_env + k)[0] for k in range(2)]
    L = ppo.PPOLearner(envs, encoder_layers=(64, 64), decoder_layers=(64, 64), critic_layers=(64, 64), latents=60, unroll_length=5,
                       batch_size=32, num_minibatches=4, num_updates_per_batch=2, seed=3, normalize_observations=True)
    for k, e in enumerate(envs):
        L.states[k] = e.reset(torch.Generator().manual_seed(10 + k))
    return L


def _learner_cfg3_rank_share():
    """One rank's share of BASELINE configs[2] (32 768 envs on 8 GPUs, the reference's batch_size 2048): 4096 envs in the bench's
    three env groups, 2 x 256 nets, 256 minibatch rows (5 120-row GEMMs), ONE unroll of 20 steps + 4 x 16 minibatch updates per training step."""
    from track_mjx_amd.agent import ppo
    sizes = ppo.group_sizes(4096, 3)
    envs = [make_env_and_oracle(num_envs=n, n_clips=4, wrappers=True, seed=0)[0] for n in sizes]
    L = ppo.PPOLearner(envs, encoder_layers=(256, 256), decoder_layers=(256, 256), critic_layers=(256, 256), latents=60, unroll_length=20,
                       batch_size=256, num_minibatches=16, num_updates_per_batch=4, seed=3, normalize_observations=True)
    assert L.unrolls == 1 and L.local_batch == 256 and L.env_steps_per_training_step == 4096 * 20
    for k, e in enumerate(envs):
        L.states[k] = e.reset(torch.Generator().manual_seed(10 + k))
    return L


@pytest.mark.gpu
def test_training_step_with_rccl_collectives_on_one_rank(monkeypatch):
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    plain = _learner(0)
    assert not plain.collectives
    for it in range(2):
        plain.training_step(it)
    torch.cuda.synchronize()
    want = plain.opt.flat.clone()
    want_mean = plain.normalizer.mean.clone()

    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        monkeypatch.setenv("TMJX_COLLECTIVES_ALWAYS", "1")
        monkeypatch.setenv("TMJX_BUCKET_OVERLAP", "1")          # (the small test nets are below the size at which the learner buckets by itself)
        ones = torch.ones(4, device=dev)
        dist.all_reduce(ones)
        assert ones.tolist() == [1.0] * 4
        L = _learner(0)
        assert L.collectives and L.world == 1 and L.use_graph
        for it in range(2):
            m = L.training_step(it)
        torch.cuda.synchronize()
        # with collectives the step runs as THREE graphs (forward + loss head | value backward | policy backward) and C1 as two buckets, the value
        # network's all-reduce issued behind ITS graph while the policy's backward graph runs (PPOLearner._capture_split)
        assert L.overlap_c1 and L._split_graphs is not None and L._graph is None, "the bucketed SGD step must run as captured graphs with RCCL initialised"
        assert all(bool(torch.isfinite(v).all()) for v in m.values())
        dp = (L.opt.flat - want).abs().max().item()
        dm = (L.normalizer.mean - want_mean).abs().max().item()
        print(f"\n[rccl one rank] max |param diff| vs plain learner {dp:.3e}, max |obs-mean diff| {dm:.3e}")
        assert dm <= 1e-6 * (1 + want_mean.abs().max().item())
        assert dp <= 1e-5, "identity collectives changed the training step"
        # the same kernels on the same data, cut into three graphs: bit-identical parameters
        assert dp == 0.0, dp
        # ... and the single-bucket path (TMJX_BUCKET_OVERLAP=0: one graph, one all-reduce of the whole buffer) likewise
        monkeypatch.setenv("TMJX_BUCKET_OVERLAP", "0")
        L1 = _learner(0)
        assert L1.collectives and not L1.overlap_c1
        for it in range(2):
            L1.training_step(it)
        torch.cuda.synchronize()
        assert L1._graph is not None and L1._split_graphs is None
        assert (L1.opt.flat - want).abs().max().item() == 0.0
        # BASELINE configs[2], one rank's share (`bench.py --config cfg3`): 64 gradient all-reduces per training step of 81 920 env-steps, each between two
        # replays of the 5 120-row SGD graph — bit-identical to the same step without collectives
        monkeypatch.delenv("TMJX_COLLECTIVES_ALWAYS")
        monkeypatch.delenv("TMJX_BUCKET_OVERLAP")
        P3 = _learner_cfg3_rank_share()
        assert not P3.collectives
        P3.training_step(0)
        torch.cuda.synchronize()
        want3 = P3.opt.flat.clone()
        del P3
        monkeypatch.setenv("TMJX_COLLECTIVES_ALWAYS", "1")
        L3 = _learner_cfg3_rank_share()
        assert L3.collectives and not L3.overlap_c1          # (2.49 MB of gradients and one rank: the single bucket)
        m3 = L3.training_step(0)
        torch.cuda.synchronize()
        assert L3._graph is not None and all(bool(torch.isfinite(v).all()) for v in m3.values())
        assert (L3.opt.flat - want3).abs().max().item() == 0.0
    finally:
        dist.barrier()
        dist.destroy_process_group()
