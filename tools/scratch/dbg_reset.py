import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.common import make_env_and_oracle
from track_mjx_amd import jax_random as jr
env,O,cl=make_env_and_oracle(num_envs=64,n_clips=4,wrappers=True)
key=jr.PRNGKey(42); st=env.reset(key); torch.cuda.synchronize()
ci,sf,qn,vn=jr.reset_draws_batch(jr.split(key,64),4,74,73,env._reset_noise_scale)
L=env.layout
qpos=env.state_buf[L.qpos:L.qpos+L.nq].cpu().numpy()
ref=np.concatenate([cl.position[ci,sf],cl.quaternion[ci,sf],cl.joints[ci,sf]],axis=-1).T
d=np.abs(qpos-(ref+qn)).max(1)
print(np.nonzero(d>1e-6)[0], d.max(), env._reset_noise_scale)
d0=np.abs(qpos-ref).max(1); print("vs no-noise", np.nonzero(d0<1e-7)[0])
