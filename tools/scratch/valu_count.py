"""VALU instruction counts of the physics kernel for different solver iteration counts (run under rocprofv3 --pmc SQ_INSTS_VALU)."""
import sys
sys.path.insert(0, '.')
import torch
from tests.common import default_walker, default_blob
from track_mjx_amd import clips as _clips
from track_mjx_amd.environment import MultiClipTracking, RewardConfig, wrap
it = int(sys.argv[1])
w, cfg = default_walker()
cl = _clips.make_synthetic_clips(w.model, 8)
ea = dict(cfg["env_config"]["env_args"]); ea["iterations"] = it; ea["ls_iterations"] = int(sys.argv[2]) if len(sys.argv) > 2 else it
env = wrap(MultiClipTracking(cl, w, RewardConfig(**cfg["env_config"]["reward_weights"]), **ea, **cfg["reference_config"], num_envs=4096, device="cuda:0"), episode_length=195)
g = torch.Generator().manual_seed(0)
st = env.reset(g)
a = (torch.randn((38, 4096), generator=g) * 0.3).clamp(-1, 1).cuda()
for _ in range(6): st = env.step(st, a)
torch.cuda.synchronize()
