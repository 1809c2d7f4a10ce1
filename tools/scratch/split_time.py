import sys, time; sys.path.insert(0,'.')
import torch
import bench as B
from track_mjx_amd import config as _config
from track_mjx_amd.agent import ppo
from track_mjx_amd.environment import wrap
from track_mjx_amd.train import build_env
device = torch.device("cuda:0"); torch.cuda.set_device(device)
cfg = _config.default_config()
cfg["network_config"].update(encoder_layer_sizes=[256, 256], decoder_layer_sizes=[256, 256], critic_layer_sizes=[256, 256])
tc = cfg["train_setup"]["train_config"]; nc = cfg["network_config"]
env = wrap(build_env(cfg, 4096, device, n_clips=64), episode_length=195)
L = ppo.PPOLearner(env, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                   latents=nc["intention_size"], learning_rate=tc["learning_rate"], entropy_cost=tc["entropy_cost"], discounting=tc["discounting"],
                   unroll_length=tc["unroll_length"], batch_size=tc["batch_size"], num_minibatches=tc["num_minibatches"],
                   num_updates_per_batch=tc["num_updates_per_batch"], normalize_observations=True, kl_weight=nc["kl_weight"], seed=0,
                   use_graph=("nograph" not in sys.argv))
g = torch.Generator().manual_seed(1); idx = torch.arange(4096, dtype=torch.int32)
L.state = env.reset(g, (idx % 64).to(torch.int32), start_frame=(idx % 44).to(torch.int32))
L.training_step(1)
for rep in range(2):
    torch.cuda.synchronize(); t0=time.perf_counter(); L.collect(); torch.cuda.synchronize(); t1=time.perf_counter(); L.update(1); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"collect {1e3*(t1-t0):.1f} ms  update {1e3*(t2-t1):.1f} ms")
# finer: act vs env.step
st = L.state
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(80): a,e = L.act(st.obs)
torch.cuda.synchronize(); t1=time.perf_counter()
for _ in range(80): st = env.step(st, a)
torch.cuda.synchronize(); t2=time.perf_counter()
print(f"80x act {1e3*(t1-t0):.1f} ms   80x env.step {1e3*(t2-t1):.1f} ms")
