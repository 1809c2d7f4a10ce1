import torch, time, os, sys
dev='cuda'
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e6
M=20480
shapes=[('fwd K696->256',M,696,256),('fwd K640->256',M,640,256),('fwd 256->256',M,256,256),('fwd 256->120',M,256,120),('fwd 256->76',M,256,76)]
for lib in ('default','cublas','cublaslt'):
    if lib!='default': torch.backends.cuda.preferred_blas_library(lib)
    print('--- blas lib', lib, torch.backends.cuda.preferred_blas_library())
    for name,m,k,n in shapes:
        x=torch.randn(m,k,device=dev); w=torch.randn(n,k,device=dev); b=torch.randn(n,device=dev); dz=torch.randn(m,n,device=dev)
        t1=bench(lambda: torch.addmm(b,x,w.t()))
        t2=bench(lambda: dz.t()@x)      # dW [n,k]
        t3=bench(lambda: dz@w)          # dx [m,k]
        fl=2*m*k*n
        print(f'{name:16s} fwd {t1:7.1f}us {fl/t1/1e6:6.1f} TF | dW {t2:7.1f}us {fl/t2/1e6:6.1f} TF | dx {t3:7.1f}us {fl/t3/1e6:6.1f} TF')
