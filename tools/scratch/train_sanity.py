import sys, time; sys.path.insert(0,'.')
import torch
from track_mjx_amd import config as _config
from track_mjx_amd.agent import ppo
from track_mjx_amd.environment import wrap
from track_mjx_amd.train import build_env
device = torch.device("cuda:0"); torch.cuda.set_device(device)
cfg = _config.default_config()
cfg["network_config"].update(encoder_layer_sizes=[256, 256], decoder_layer_sizes=[256, 256], critic_layer_sizes=[256, 256])
tc = cfg["train_setup"]["train_config"]; nc = cfg["network_config"]
envs = [wrap(build_env(cfg, 2048, device, n_clips=64), episode_length=195) for _ in range(2)]
L = ppo.PPOLearner(envs, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                   latents=nc["intention_size"], learning_rate=tc["learning_rate"], entropy_cost=tc["entropy_cost"], discounting=tc["discounting"],
                   unroll_length=tc["unroll_length"], batch_size=tc["batch_size"], num_minibatches=tc["num_minibatches"],
                   num_updates_per_batch=tc["num_updates_per_batch"], normalize_observations=True, kl_weight=nc["kl_weight"], seed=0)
g = torch.Generator().manual_seed(1)
for k,e in enumerate(envs): L.states[k] = e.reset(g)
for it in range(int(sys.argv[1]) if len(sys.argv)>1 else 12):
    m = L.training_step(1)
    rew = L.buf["reward"].mean().item(); done = (1-L.buf["discount"]).mean().item()
    print(it, {k: round(float(v),4) for k,v in m.items()}, 'mean reward', round(rew,4), 'done frac', round(done,4), flush=True)
