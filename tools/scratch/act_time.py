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
env = wrap(build_env(cfg, 2048, device, n_clips=64), episode_length=195)
L = ppo.PPOLearner(env, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                   latents=nc["intention_size"], unroll_length=20, batch_size=512, num_minibatches=16, num_updates_per_batch=4, normalize_observations=True, seed=0)
g = torch.Generator().manual_seed(1)
st = env.reset(g)
for lf in (False, True):
    L.lds_free = lf; L._act_graphs = {}
    for _ in range(3): L._act_graphed(st.obs, 0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): L._act_graphed(st.obs, 0)
    torch.cuda.synchronize(); print('lds_free',lf,'act ms', (time.perf_counter()-t)/50*1e3)
