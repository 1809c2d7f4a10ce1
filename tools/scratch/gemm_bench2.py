import torch, time
dev='cuda'
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e6
M=20480
for name,k,n in [('K696->256',696,256),('K640->256',640,256),('256->256',256,256),('256->120',256,120),('256->76',256,76),('376->256',321,256)]:
    x=torch.randn(M,k,device=dev); dz=torch.randn(M,n,device=dev)
    ref=dz.t()@x
    t0=bench(lambda: dz.t()@x)
    out=[f'{name:10s} plain {t0:6.1f}us']
    for S in (4,8,16,32):
        def f():
            return torch.bmm(dz.view(S,M//S,n).transpose(1,2), x.view(S,M//S,k)).sum(0)
        err=(f()-ref).abs().max().item()/ref.abs().max().item()
        out.append(f'S{S} {bench(f):6.1f}us')
    # baddbmm-free alternative: einsum
    print(' | '.join(out), f'err {err:.1e}')
