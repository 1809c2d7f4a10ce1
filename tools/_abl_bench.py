import sys, ctypes as C, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from track_mjx_amd import hip
from tools.gemm_bench import timeit
DEV = "cuda:0"
M = 20480
p = lambda t: C.c_void_p(t.data_ptr())
for name in sys.argv[1:]:
    L = hip.load(Path(name))
    out = []
    for N, K in [(256, 256), (256, 472), (256, 1024), (512, 1024)]:
        x = torch.randn((M, K), device=DEV); w = torch.randn((N, K), device=DEV); z = torch.empty((M, N), device=DEV); dy = torch.randn((M, N), device=DEV); dx = torch.empty((M, K), device=DEV)
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        t = timeit(lambda: L.tmjx_gemm_nt(p(x), K, p(w), K, None, p(z), N, M, N, K, s), n=30)
        t2 = timeit(lambda: L.tmjx_gemm_nn(p(dy), N, p(w), K, p(dx), K, M, K, N, s), n=30)
        out.append(f"N={N} K={K}: nt {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:6.1f} TF nn {t2*1e6:6.1f} us {2.0*M*N*K/t2/1e12:6.1f} TF")
    print(f"{name}: " + " | ".join(out), flush=True)
