import sys, time; sys.path.insert(0,'.')
import torch
from track_mjx_amd.agent.networks import _SiluLayerNormFn
dev=torch.device('cuda:0')
z=torch.randn(20480,256,device=dev,requires_grad=True); b=torch.randn(256,device=dev,requires_grad=True); g=torch.ones(256,device=dev,requires_grad=True); be=torch.zeros(256,device=dev,requires_grad=True)
up=torch.randn(20480,256,device=dev)
def f():
    y=_SiluLayerNormFn.apply(z,b,g,be,1e-6); return torch.autograd.grad(y,(z,b,g,be),up)
for _ in range(5): f()
torch.cuda.synchronize(); t=time.time()
for _ in range(50): f()
torch.cuda.synchronize(); print('fwd+bwd us',(time.time()-t)/50*1e6)
