"""Time tmjx_linear_nolds alone (2048 x 256 x 256 and the 470-wide first layer) and check it against torch."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import torch
from track_mjx_amd import hip as _hip
L = _hip.lib()
dev = torch.device('cuda:0')
for (M, N, K, kmajor) in ((2048, 256, 256, False), (2048, 256, 472, True), (2048, 120, 256, False), (2048, 76, 256, False), (132, 64, 36, False), (2048, 1024, 472, True), (2048, 512, 1024, False), (100, 40, 20, True)):
    A = torch.randn((K, M), device=dev).t() if kmajor else torch.randn((M, K), device=dev)
    W = torch.randn((N, K), device=dev); b = torch.randn(N, device=dev); out = torch.empty((M, N), device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    def run():
        _hip.check(L.tmjx_linear_nolds(p(A), A.stride(0), A.stride(1), p(W), p(b), p(out), M, N, K, s), "nolds")
    run(); torch.cuda.synchronize()
    ref = (A.double() @ W.double().t() + b.double())
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    t0 = time.time()
    for _ in range(200): run()
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K} kmajor={kmajor}: {(time.time() - t0) / 200 * 1e6:.1f} us  rel err {err:.2e}")
