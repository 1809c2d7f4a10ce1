"""Is the roll-out loop CPU-bound?  Time collect() enqueue (no sync) vs completion, and update() likewise."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench as B
from track_mjx_amd import config as _config, clips as _clips
from track_mjx_amd.agent import ppo
from track_mjx_amd.environment import wrap
from track_mjx_amd.train import build_env
from track_mjx_amd.walker import Rodent
dev = torch.device('cuda:0')
cfg = _config.default_config(); cfg["network_config"].update(**B.CONFIGS["cfg2"]["nets"])
tc, nc = cfg["train_setup"]["train_config"], cfg["network_config"]
table = _clips.make_synthetic_clips(Rodent(**cfg["walker_config"]).model, 64)
ngrp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
envs = [wrap(build_env(cfg, 4096 // ngrp, dev, reference_clip=table), episode_length=195) for _ in range(ngrp)]
L = ppo.PPOLearner(envs if ngrp > 1 else envs[0], encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                   latents=60, unroll_length=20, batch_size=1024, num_minibatches=16, num_updates_per_batch=4, kl_weight=nc["kl_weight"], seed=0)
g = torch.Generator().manual_seed(1)
for k, e in enumerate(envs): L.states[k] = e.reset(g)
L.training_step(1); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); L.collect(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    L.update(1); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"groups={ngrp} collect: enqueue {1e3*(t1-t0):.1f} ms, done {1e3*(t2-t0):.1f} ms | update: enqueue {1e3*(t3-t2):.1f} ms, done {1e3*(t4-t2):.1f} ms", flush=True)
