import sys, time; sys.path.insert(0,'.')
import torch
from tests.common import make_env_and_oracle
dev=torch.device('cuda:0')
def mk(n):
    env,_,_ = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
    g=torch.Generator().manual_seed(0); st=env.reset(g)
    return env, st
acts={n:[(torch.randn((38,n))*0.3).clamp(-1,1).cuda() for _ in range(4)] for n in (2048,4096)}
envA,stA=mk(4096)
for i in range(4): stA=envA.step(stA,acts[4096][i%4])
torch.cuda.synchronize(); t=time.time()
for i in range(40): stA=envA.step(stA,acts[4096][i%4])
torch.cuda.synchronize(); print('one env 4096: ms/step',(time.time()-t)/40*1e3)
e0,s0=mk(2048); e1,s1=mk(2048)
sa,sb=torch.cuda.Stream(),torch.cuda.Stream()
# extra elementwise work emulating act (a few small matmuls) per group
W=[torch.randn(696,256,device=dev),torch.randn(256,256,device=dev),torch.randn(256,76,device=dev)]
def act_like(obs):
    x=(obs if obs.shape[-1]==696 else obs.t())@W[0]; x=torch.nn.functional.silu(x)@W[1]; x=torch.nn.functional.silu(x)@W[2]; return torch.tanh(x[:,:38]).t().contiguous()
for rep in range(2):
    torch.cuda.synchronize(); t=time.time()
    for i in range(40):
        with torch.cuda.stream(sa):
            a=act_like(s0.obs) if rep else acts[2048][i%4]; s0=e0.step(s0,a)
        with torch.cuda.stream(sb):
            a=act_like(s1.obs) if rep else acts[2048][i%4]; s1=e1.step(s1,a)
    torch.cuda.synchronize(); print(('with act-like work' if rep else 'physics only'),'two envs 2048 on two streams: ms/step',(time.time()-t)/40*1e3)
# single stream with act-like
torch.cuda.synchronize(); t=time.time()
for i in range(40):
    a=act_like(stA.obs); stA=envA.step(stA,a)
torch.cuda.synchronize(); print('one env 4096 with act-like: ms/step',(time.time()-t)/40*1e3)
