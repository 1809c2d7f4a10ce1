"""CPU: learner-side maths against independent formulations (the reference's own modules are not importable here)."""
import math

import numpy as np
import torch

from track_mjx_amd import config as _config
from track_mjx_amd.agent import losses
from track_mjx_amd.agent.networks import IntentionPolicy, NormalTanh, RunningStatistics, ValueNet


def _gae_torch(trunc, term, rew, val, boot, lam, disc):
    """plain fp32 torch reference of losses.py:39-100 (reverse python loop)"""
    T = rew.shape[0]
    tm = 1 - trunc
    v1 = torch.cat([val[1:], boot[None]], 0)
    deltas = (rew + disc * (1 - term) * v1 - val) * tm
    acc = torch.zeros_like(boot); out = []
    for t in range(T - 1, -1, -1):
        acc = deltas[t] + disc * (1 - term[t]) * tm[t] * lam * acc
        out.append(acc)
    vs = torch.stack(out[::-1]) + val
    adv = (rew + disc * (1 - term) * torch.cat([vs[1:], boot[None]], 0) - val) * tm
    return vs, adv


def test_normal_tanh_matches_torch_distributions():
    torch.manual_seed(0)
    logits = torch.randn(5, 76)
    raw = torch.randn(5, 38)
    loc, scale = NormalTanh.params(logits)
    base = torch.distributions.Normal(loc, scale)
    td = torch.distributions.TransformedDistribution(base, [torch.distributions.transforms.TanhTransform(cache_size=1)])
    lp_ref = td.log_prob(torch.tanh(raw).clamp(-1 + 1e-7, 1 - 1e-7)).sum(-1)
    assert torch.allclose(NormalTanh.log_prob(logits, raw), lp_ref, atol=2e-3)
    assert (scale > 0.001).all() and torch.allclose(NormalTanh.mode(logits), torch.tanh(loc))


def test_running_statistics_two_batches():
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(100, 7)).astype(np.float32) * 3 + 1, rng.normal(size=(50, 7)).astype(np.float32)
    rs = RunningStatistics(7, "cpu")
    rs.update(torch.from_numpy(a)); rs.update(torch.from_numpy(b))
    allx = np.concatenate([a, b])
    np.testing.assert_allclose(rs.mean.numpy(), allx.mean(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rs.std.numpy(), allx.std(0), rtol=1e-4)
    rs2 = RunningStatistics(3, "cpu"); rs2.update(torch.zeros(10, 3))
    assert (rs2.std == 1e-6).all()                       # std clip floor (masked_running_statistics.py:203-206)


def test_network_shapes_and_parameter_counts():
    cfg = _config.default_config()["network_config"]
    pol = IntentionPolicy(696, 470, 38, 60, [256, 256], [256, 256]); val = ValueNet(696, [256, 256])
    n = sum(p.numel() for p in pol.parameters()) + sum(p.numel() for p in val.parameters())
    assert n == 622533                                     # 2x256 nets
    pol2 = IntentionPolicy(696, 470, 38, 60, cfg["encoder_layer_sizes"], cfg["decoder_layer_sizes"]); val2 = ValueNet(696, cfg["critic_layer_sizes"])
    assert sum(p.numel() for p in pol2.parameters()) + sum(p.numel() for p in val2.parameters()) == 4294853
    logits, mu, lv = pol(torch.randn(3, 696))
    assert logits.shape == (3, 76) and mu.shape == (3, 60) and val(torch.randn(3, 696)).shape == (3,)
    # LayerNorm comes after the activation (intention_network.py:38-39): the block output is normalised
    h = pol.encoder[0](torch.randn(4, 470))
    assert torch.allclose(h.mean(-1), torch.zeros(4), atol=1e-5)


def test_ppo_loss_pieces_with_torch_gae():
    torch.manual_seed(0)
    T, B = 5, 6
    pol = IntentionPolicy(30, 20, 4, 8, [16], [16]); val = ValueNet(30, [16]); rs = RunningStatistics(30, "cpu")
    data = {"observation": torch.randn(T, B, 30), "next_observation_last": torch.randn(B, 30), "reward": torch.randn(T, B),
            "discount": (torch.rand(T, B) > 0.1).float(), "truncation": (torch.rand(T, B) > 0.9).float(),
            "raw_action": torch.randn(T, B, 4), "log_prob": torch.randn(T, B) - 4}
    total, m = losses.compute_ppo_loss(pol, val, rs, data, entropy_cost=1e-2, kl_weight=0.1, discounting=0.98, clipping_epsilon=0.2, gae_fn=_gae_torch)
    assert torch.isfinite(total)
    assert abs(float(m["total_loss"]) - float(m["policy_loss"] + m["v_loss"] + m["entropy_loss"] + m["kl_latent_loss"])) < 1e-5
    total.backward()
    assert all(p.grad is not None for p in pol.parameters())
    # KL term by hand for T = 2 (losses.py:200-235)
    mu = torch.tensor([[[0.5]], [[0.2]]]); lv = torch.tensor([[[0.1]], [[-0.3]]])
    kl0 = -0.5 * (1 + 0.1 - 0.25 - math.exp(0.1))
    pv = 1 - 0.95 ** 2
    klt = 0.5 * (math.exp(-0.3) / pv + (0.95 * 0.5 - 0.2) ** 2 / pv - 1 + math.log(pv) + 0.3)
    expect = 0.1 * ((kl0 + klt * 1) / 2)

    class P(torch.nn.Module):
        def forward(self, obs):
            return torch.zeros(2, 1, 8), mu, lv
    d2 = {"observation": torch.zeros(2, 1, 30), "next_observation_last": torch.zeros(1, 30), "reward": torch.zeros(2, 1), "discount": torch.ones(2, 1),
          "truncation": torch.zeros(2, 1), "raw_action": torch.zeros(2, 1, 4), "log_prob": torch.zeros(2, 1)}
    _, m2 = losses.compute_ppo_loss(P(), val, rs, d2, kl_weight=0.1, gae_fn=_gae_torch)
    assert abs(float(m2["kl_latent_loss"]) - expect) < 1e-6
    sch = losses.create_ramp_schedule(max_value=0.1, ramp_steps=37)
    assert abs(sch(0) - 1e-5) < 1e-12 and abs(sch(18.5) - 0.05) < 1e-9 and sch(100) == 0.1


def test_bf16_mode_has_no_library_gemm_path():
    """BASELINE config 5: the bf16 GEMM-input mode must go through the library's own kernels (csrc/gemm_bf16.h) — the torch.mm / bmm paths of rounds
    1-2 (and their operand casts) are gone from the product source."""
    import re
    from pathlib import Path
    src = (Path(__file__).resolve().parents[1] / "track_mjx_amd" / "agent" / "networks.py").read_text()
    assert not re.search(r"torch\.(mm|bmm|addmm|matmul)\(", src) and "out_dtype" not in src
    assert "tmjx_bgemm_nt" in src and "tmjx_bgemm_dw" in src and "tmjx_bgemm_ln_bwd" in src


def _flat_adam_vs_torch(device, steps=5, tol=1e-6):
    from track_mjx_amd.agent.ppo import FlatAdam, FlatGrads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(device)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(device)
    ref.load_state_dict(net.state_dict())
    fg = FlatGrads(list(net.parameters()))
    opt = FlatAdam(fg, 1e-2, max_norm=0.5)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(1)
    for s in range(steps):
        x = torch.randn(16, 7, generator=g).to(device) * (10.0 if s % 2 else 0.01)     # norms on both sides of the clip threshold
        fg.assign(torch.autograd.grad(net(x).square().sum(), fg.params))
        opt.step()
        ropt.zero_grad()
        ref(x).square().sum().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)       # same scale: max_norm / max(norm, max_norm) up to its 1e-6 epsilon
        ropt.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=tol), float((a - b).abs().max())
    # parameters are views of one flat buffer and stay so after the steps
    assert all(p.data_ptr() >= opt.flat.data_ptr() and p.data_ptr() < opt.flat.data_ptr() + opt.flat.numel() * 4 for p in net.parameters())


def test_flat_adam_matches_torch_adam_with_global_norm_clip():
    _flat_adam_vs_torch("cpu")


def test_flax_named_checkpoint_round_trip(tmp_path):
    """agent/checkpoint.py: the policy exported in the reference's flax tree naming ([in, out] kernels, fc2_mean / fc2_logvar,
    the decoder's last hidden_L = output layer) and loaded back into a differently initialised policy gives the same outputs."""
    from track_mjx_amd.agent import checkpoint as ck
    from track_mjx_amd.agent.networks import IntentionPolicy, RunningStatistics
    torch.manual_seed(0)
    a = IntentionPolicy(696, 470, 38, 60, (64, 32), (48, 24))
    tree = ck.policy_to_flax(a)
    p = tree["params"]
    assert sorted(p["encoder"]) == ["LayerNorm_0", "LayerNorm_1", "fc2_logvar", "fc2_mean", "hidden_0", "hidden_1"]
    assert sorted(p["decoder"]) == ["LayerNorm_0", "LayerNorm_1", "hidden_0", "hidden_1", "hidden_2"]
    assert p["encoder"]["hidden_0"]["kernel"].shape == (470, 64) and p["encoder"]["fc2_mean"]["kernel"].shape == (32, 60)
    assert p["decoder"]["hidden_0"]["kernel"].shape == (60 + 226, 48) and p["decoder"]["hidden_2"]["kernel"].shape == (24, 76)
    flat = ck.flatten({"policy": tree})
    assert "policy/params/encoder/hidden_0/kernel" in flat
    torch.manual_seed(1)
    b = IntentionPolicy(696, 470, 38, 60, (64, 32), (48, 24))
    ck.policy_from_flax(b, ck.unflatten(flat)["policy"])
    x, eps = torch.randn(5, 696), torch.randn(5, 60)
    for u, v in zip(a(x, eps=eps), b(x, eps=eps)):
        assert torch.equal(u, v)
    n1, n2 = RunningStatistics(696, "cpu"), RunningStatistics(696, "cpu")
    n1.update(torch.randn(50, 696))
    ck.normalizer_from_flax(n2, ck.normalizer_to_flax(n1))
    assert torch.equal(n1.std, n2.std) and float(n2.count) == 50.0


def test_full_learner_checkpoint_round_trip(tmp_path):
    """save_npz / load_npz on a whole learner: normaliser, policy, VALUE network, Adam moments and step count come back (a resumed
    run must not pair a trained policy with a fresh critic and zero optimiser state), in place (flat-buffer views stay valid)."""
    from tests.common import StubEnv, torch_gae
    from track_mjx_amd.agent import checkpoint as ck
    from track_mjx_amd.agent.ppo import PPOLearner

    def make(seed):
        ln = PPOLearner(StubEnv(4, 24, 16, 3), encoder_layers=(12,), decoder_layers=(10,), critic_layers=(8,), latents=4, unroll_length=3,
                        batch_size=4, num_minibatches=2, num_updates_per_batch=1, learning_rate=1e-2, use_graph=False, seed=seed)
        ln.gae_fn = torch_gae
        return ln
    a = make(3)
    g = torch.Generator().manual_seed(0)
    for k, v in a.buf.items():
        v.copy_(torch.rand(v.shape, generator=g))
    a.update()
    path = tmp_path / "PPONetwork_0.npz"
    ck.save_npz(path, a, config={"train_setup": {"seed": 3}}, step=123)
    b = make(4)
    ptrs = [p.data_ptr() for p in b.params]
    extra = ck.load_npz(path, b)
    assert extra["env_steps"] == 123 and extra["config"]["train_setup"]["seed"] == 3
    assert torch.equal(a.opt.flat, b.opt.flat) and torch.equal(a.opt.exp_avg, b.opt.exp_avg) and torch.equal(a.opt.exp_avg_sq, b.opt.exp_avg_sq)
    assert a.opt.t == b.opt.t == 2
    assert torch.equal(a.normalizer.mean, b.normalizer.mean) and float(b.normalizer.count) == float(a.normalizer.count)
    x = torch.randn(5, 24)
    assert torch.equal(a.value(x), b.value(x))
    assert [p.data_ptr() for p in b.params] == ptrs          # loaded in place
    assert sorted(k for k in ck.flatten(ck.learner_tree(a)) if k.startswith("value/")) == [
        "value/params/hidden_0/bias", "value/params/hidden_0/kernel", "value/params/hidden_1/bias", "value/params/hidden_1/kernel"]


def test_flat_buffers_pad_odd_row_lengths_and_keep_pads_zero():
    """Weight matrices whose row length is not a multiple of 4 (the 470- / 286-wide first layers) live in the flat buffers with rows padded
    to one (16-byte aligned rows for the GEMM kernels' vector loads): the parameter is a strided view, the pad stays exactly zero through
    optimiser steps, and the step equals torch's Adam with global-norm clipping on the unpadded tensors."""
    from track_mjx_amd.agent.ppo import FlatAdam, FlatGrads
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.SiLU(), torch.nn.Linear(5, 3))
    ref.load_state_dict(net.state_dict())
    fg = FlatGrads(list(net.parameters()))
    opt = FlatAdam(fg, lr=1e-2, max_norm=0.05)
    w0 = net[0].weight
    assert w0.shape == (5, 7) and w0.stride() == (8, 1) and w0.data_ptr() % 16 == 0 and net[2].weight.stride() == (8, 1)
    assert all(p.data_ptr() % 16 == 0 for p in net.parameters())
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, eps=1e-8)
    x = torch.randn(11, 7)
    for _ in range(3):
        fg.assign(torch.autograd.grad((net(x) ** 2).sum(), fg.params))
        opt.step()
        ropt.zero_grad()
        (ref(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
        ropt.step()
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    pad = opt.flat[:40].view(5, 8)[:, 7]
    assert (pad == 0).all() and (opt.exp_avg[:40].view(5, 8)[:, 7] == 0).all()


def test_checkpoint_step_directory_is_atomic_named_like_the_reference_and_never_overwritten(tmp_path):
    """save_step_dir / load_step_dir: <dir>/<it>/{policy.npz, train_state.npz, config/metadata} — the item names of the reference's
    Composite save (checkpointing.py:280-306), optimiser moments as per-parameter trees in the parameters' own naming (independent of the
    flat layout's pad rule), env_steps / iteration / noise-stream positions restored, an existing step refused, no temporary left behind."""
    import json
    import os
    import numpy as np
    import pytest
    from tests.common import StubEnv, torch_gae
    from track_mjx_amd.agent import checkpoint as ck
    from track_mjx_amd.agent.ppo import PPOLearner

    def make(seed):
        ln = PPOLearner(StubEnv(4, 24, 16, 3), encoder_layers=(12,), decoder_layers=(10,), critic_layers=(8,), latents=4, unroll_length=3,
                        batch_size=4, num_minibatches=2, num_updates_per_batch=1, learning_rate=1e-2, use_graph=False, seed=seed)
        ln.gae_fn = torch_gae
        return ln
    a = make(3)
    g = torch.Generator().manual_seed(0)
    for k, v in a.buf.items():
        v.copy_(torch.rand(v.shape, generator=g))
    a.update()
    a._mb_state[0] = 77                                     # the SGD step's Philox draw counter
    a._act_rng_state(a.gens[0])[0][0] = 1234                # the acting stream's counter
    d = tmp_path / "ckpt"
    final = ck.save_step_dir(d, 2, a, config={"train_setup": {"seed": 3}}, env_steps=655360)
    assert sorted(os.listdir(final)) == ["config", "policy.npz", "train_state.npz"] and os.listdir(d) == ["2"]
    assert json.load(open(os.path.join(final, "config", "metadata")))["train_setup"]["seed"] == 3
    with np.load(os.path.join(final, "policy.npz")) as z:
        assert "0/mean" in z.files and "1/params/encoder/hidden_0/kernel" in z.files and z["1/params/encoder/hidden_0/kernel"].shape == (16, 12)
    with np.load(os.path.join(final, "train_state.npz")) as z:
        # un-padded per-parameter moments: the 16-wide first layer is stored [in, out] like its parameter
        assert z["optimizer_state/mu/policy/params/encoder/hidden_0/kernel"].shape == (16, 12)
        assert z["optimizer_state/nu/value/params/hidden_0/kernel"].shape == (24, 8) and int(z["env_steps"]) == 655360
    with pytest.raises(FileExistsError):
        ck.save_step_dir(d, 2, a)
    assert ck.latest_step(d) == 2 and ck.latest_step(tmp_path / "nothing") is None
    b = make(4)
    extra = ck.restore(d, b)                                # the directory: its latest step
    assert extra == {"config": {"train_setup": {"seed": 3}}, "env_steps": 655360, "iteration": 2}
    assert torch.equal(a.opt.flat, b.opt.flat) and torch.equal(a.opt.exp_avg, b.opt.exp_avg) and torch.equal(a.opt.exp_avg_sq, b.opt.exp_avg_sq)
    assert b.opt.t == a.opt.t and int(b._mb_state[0]) == 77 and int(b._act_rng_state(b.gens[0])[0][0]) == 1234
    assert torch.equal(a.gens[0].get_state(), b.gens[0].get_state())
    # .npz form: refuses to overwrite too, and carries the iteration
    ck.save_npz(tmp_path / "x.npz", a, step=5, iteration=1)
    with pytest.raises(FileExistsError):
        ck.save_npz(tmp_path / "x.npz", a, step=5)
    assert ck.restore(tmp_path / "x.npz", make(5))["iteration"] == 1
    assert not [f for f in os.listdir(tmp_path) if f.startswith(".tmp")]


def test_checkpointed_generator_states_do_not_collapse_the_ranks():
    """agent/checkpoint.py: rank 0 writes the noise-stream positions; at resume every rank restores from that one tree.  The saving rank must
    continue its torch generators exactly, every OTHER rank must get a stream of its own (derived from the saved state and its rank) — not a
    copy of rank 0's (round-3 advisor finding: all ranks then drew identical shuffles / noise)."""
    import torch
    from track_mjx_amd.agent import checkpoint as ck

    class Stub:
        def __init__(self, rank):
            self.rank = rank
            self._mb_state = torch.zeros(16, dtype=torch.long)
            self.gens = [torch.Generator().manual_seed(1000 + 17 + rank), torch.Generator().manual_seed(1000 + 17 + rank + 7919)]
            self._act_rng = {}

        def _act_rng_state(self, gen):
            return self._act_rng.setdefault(id(gen), (torch.zeros(2, dtype=torch.long), 0))

    saver = Stub(0)
    torch.randn(5, generator=saver.gens[0])          # the run has advanced
    saver._mb_state[0] = 77
    tree = ck.rng_tree(saver)
    want = [torch.randn(4, generator=g) for g in saver.gens]          # what rank 0 draws next
    r0, r1, r2 = Stub(0), Stub(1), Stub(2)
    for r in (r0, r1, r2):
        ck.rng_from_tree(r, tree)
        assert int(r._mb_state[0]) == 77
    got = {r.rank: [torch.randn(4, generator=g) for g in r.gens] for r in (r0, r1, r2)}
    assert all(torch.equal(a, b) for a, b in zip(got[0], want))
    assert not torch.equal(got[1][0], got[0][0]) and not torch.equal(got[2][0], got[0][0]) and not torch.equal(got[1][0], got[2][0])
    assert not torch.equal(got[1][0], got[1][1])
    # deterministic: the same (tree, rank) gives the same stream again
    r1b = Stub(1)
    ck.rng_from_tree(r1b, tree)
    assert torch.equal(torch.randn(4, generator=r1b.gens[0]), got[1][0])
