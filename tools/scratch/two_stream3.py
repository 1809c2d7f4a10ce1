import sys, time; sys.path.insert(0,'.')
import torch
from tests.common import make_env_and_oracle
dev=torch.device('cuda:0')
def mk(n):
    env,_,_ = make_env_and_oracle(num_envs=n, n_clips=64, wrappers=True)
    g=torch.Generator().manual_seed(0); st=env.reset(g)
    return env, st
e0,s0=mk(2048); e1,s1=mk(2048)
sa,sb=torch.cuda.Stream(),torch.cuda.Stream()
big=[torch.randn(2048,4096,device=dev) for _ in range(2)]
def fake_act(obs, buf, reps):
    # LDS-free elementwise work standing in for the policy inference
    x=buf
    for _ in range(reps): x=torch.sin(x)*1.0001+0.1
    a=torch.tanh((obs if obs.shape[0]==696 else obs.t())[:38]*0.01 + x[:38,:2048]*0.0)
    return a.contiguous()
for reps in (0,4,12):
    for i in range(3):
        with torch.cuda.stream(sa): s0=e0.step(s0,fake_act(s0.obs,big[0],reps))
        with torch.cuda.stream(sb): s1=e1.step(s1,fake_act(s1.obs,big[1],reps))
    torch.cuda.synchronize(); t=time.time()
    for i in range(40):
        with torch.cuda.stream(sa): s0=e0.step(s0,fake_act(s0.obs,big[0],reps))
        with torch.cuda.stream(sb): s1=e1.step(s1,fake_act(s1.obs,big[1],reps))
    torch.cuda.synchronize(); dt=(time.time()-t)/40*1e3
    # standalone cost of the fake act
    torch.cuda.synchronize(); t=time.time()
    for i in range(40): fake_act(s0.obs,big[0],reps)
    torch.cuda.synchronize(); da=(time.time()-t)/40*1e3
    print(f'reps={reps}: two streams {dt:.2f} ms per step pair; fake act alone {da:.3f} ms per half')
envA,stA=mk(4096)
def act0(obs): return torch.tanh((obs if obs.shape[0]==696 else obs.t())[:38]*0.01).contiguous()
for i in range(3): stA=envA.step(stA,act0(stA.obs))
torch.cuda.synchronize(); t=time.time()
for i in range(40): stA=envA.step(stA,act0(stA.obs))
torch.cuda.synchronize(); print(f'single env 4096, same kind of actions: {(time.time()-t)/40*1e3:.2f} ms per step')
