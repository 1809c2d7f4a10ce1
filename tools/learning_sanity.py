#!/usr/bin/env python3
"""Learning sanity of the whole path (physics kernel, K3, hipGraph inference, self-advancing SGD step with the hand-written GEMMs): the bench
configuration (4096 envs, 2x256 nets, synthetic clips, random init) trained for N steps; prints the mean per-step roll-out reward, the
losses and the NaN-guard rate every few steps.  GPU box: python tools/learning_sanity.py [steps=80] [bf16 | cfg4 | cfg5]   (cfg4 / cfg5: that BASELINE
configuration's env count, clip table and nets as bench.py --config builds them; bf16: the MLP GEMMs in bf16 GEMM-input mode,
BASELINE config 5's numerics — bf16 operands, bf16 saved pre-activations — on the same configuration, for a like-for-like reward curve)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from track_mjx_amd import config as _config  # noqa: E402
from track_mjx_amd.agent import ppo  # noqa: E402
from track_mjx_amd.environment import wrap  # noqa: E402
from track_mjx_amd.train import build_env  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
mode = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
bf16 = mode in ("bf16", "cfg5")
dev = torch.device("cuda:0")
cfg = _config.default_config()
import bench as _bench  # noqa: E402
bc = _bench.CONFIGS[mode if mode in _bench.CONFIGS else "cfg2"]
cfg["network_config"].update(**bc["nets"])
tc, nc = cfg["train_setup"]["train_config"], cfg["network_config"]
n_envs = bc["envs_per_gpu"]
sizes = ppo.group_sizes(n_envs, ppo.default_groups(n_envs, dev))           # the bench's env groups (three since round 3)
envs = [wrap(build_env(cfg, n, dev, n_clips=min(bc["n_clips"], 64)), episode_length=195) for n in sizes]
L = ppo.PPOLearner(envs, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                   latents=nc["intention_size"], learning_rate=tc["learning_rate"], entropy_cost=tc["entropy_cost"], discounting=tc["discounting"],
                   unroll_length=tc["unroll_length"], batch_size=tc["batch_size"] * n_envs // 4096, num_minibatches=tc["num_minibatches"],
                   num_updates_per_batch=tc["num_updates_per_batch"], normalize_observations=True, kl_weight=nc["kl_weight"], seed=0,
                   matmul_dtype=torch.bfloat16 if bf16 else None)
g = torch.Generator().manual_seed(1)
for k, e in enumerate(envs):
    L.states[k] = e.reset(g)
t0 = time.perf_counter()
for it in range(steps):
    m = L.training_step(it)
    if it % 8 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        r = L.buf["reward"]
        ep_end = 1.0 - L.buf["discount"].mean().item()
        print(f"step {it:3d}  env-steps {(it + 1) * L.env_steps_per_training_step / 1e6:6.1f} M  mean reward/step {r.mean().item():+.3f}  episode ends/step {ep_end:.3f}  "
              f"total {m['total_loss'].item():+.3f}  policy {m['policy_loss'].item():+.4f}  value {m['v_loss'].item():.4f}  kl {m['kl_latent_loss'].item():.4f}  "
              f"entropy {m['entropy_loss'].item():+.4f}  finite {bool(torch.isfinite(r).all())}  {time.perf_counter() - t0:5.1f} s  "
              f"mem {torch.cuda.memory_allocated() / 2 ** 20:.0f} MiB (peak {torch.cuda.max_memory_allocated() / 2 ** 20:.0f})", flush=True)
