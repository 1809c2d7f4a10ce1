import sys; sys.path.insert(0,'.')
import torch
from track_mjx_amd.agent import losses
from track_mjx_amd.agent.networks import IntentionPolicy, RunningStatistics, ValueNet, NormalTanh
dev = torch.device("cuda:0")
torch.manual_seed(5)
obs, ref, nu, Z, T, B = 96, 40, 38, 60, 7, 96
policy = IntentionPolicy(obs, ref, nu, Z, (64, 64), (64, 64)).to(dev)
value = ValueNet(obs, (64, 64)).to(dev)
with torch.no_grad(): policy.head.weight.mul_(0.05)
norm = RunningStatistics(obs, dev)
g = torch.Generator(device=dev).manual_seed(11)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
data = {"observation": rnd(T, B, obs), "next_observation_last": rnd(B, obs), "raw_action": rnd(T, B, nu) * 0.7,
        "log_prob": rnd(T, B) * 0.3 - 20.0, "reward": rnd(T, B).abs(),
        "discount": (torch.rand(T, B, generator=g, device=dev) > 0.1).float(),
        "truncation": (torch.rand(T, B, generator=g, device=dev) > 0.9).float()}
with torch.no_grad():
    lg, _, _ = policy(norm.normalize(data["observation"]))
    data["log_prob"] = NormalTanh.log_prob(lg, data["raw_action"]) + rnd(T, B) * 0.3
hp = dict(entropy_cost=1e-2, kl_weight=0.1, discounting=0.95, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.2)
torch.manual_seed(99)
l_ref, m_ref = losses.compute_ppo_loss(policy, value, norm, data, **hp)
torch.manual_seed(99)
l_fus, m_fus = losses.compute_ppo_loss_fused(policy, value, norm, data, **hp)
print({k: float(v) for k, v in m_ref.items()})
print({k: float(v) for k, v in m_fus.items()})
print(float(norm.std.min()), float(norm.count))
