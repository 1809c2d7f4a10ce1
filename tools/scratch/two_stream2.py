import sys, time; sys.path.insert(0,'.')
import torch
dev=torch.device('cuda:0')
W=[torch.randn(696,256,device=dev),torch.randn(256,256,device=dev),torch.randn(256,76,device=dev)]
obs=[torch.randn(2048,696,device=dev) for _ in range(2)]
def act_like(o):
    x=o@W[0]; x=torch.nn.functional.silu(x)@W[1]; x=torch.nn.functional.silu(x)@W[2]; return torch.tanh(x[:,:38]).t().contiguous()
sa,sb=torch.cuda.Stream(),torch.cuda.Stream()
for mode in ('default','streams','streams'):
    torch.cuda.synchronize(); t=time.time()
    for i in range(100):
        if mode=='default':
            act_like(obs[0]); act_like(obs[1])
        else:
            with torch.cuda.stream(sa): act_like(obs[0])
            with torch.cuda.stream(sb): act_like(obs[1])
    torch.cuda.synchronize(); print(mode,(time.time()-t)/100*1e3,'ms per pair')
