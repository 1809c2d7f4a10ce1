import torch, time
dev = 'cuda:0'
M, K, N = 40960, 512, 512
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
xb, wb = x.bfloat16(), w.bfloat16()
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e6
print("fp32 mm        %.1f us" % t(lambda: x @ w.t()))
print("bf16 mm        %.1f us" % t(lambda: xb @ wb.t()))
try:
    y = torch.mm(xb, wb.t(), out_dtype=torch.float32)
    print("bf16->f32 mm   %.1f us" % t(lambda: torch.mm(xb, wb.t(), out_dtype=torch.float32)), y.dtype, float((y - x @ w.t()).abs().max() / (x @ w.t()).abs().max()))
except Exception as e:
    print("out_dtype unsupported:", type(e).__name__, str(e)[:200])
print("cast f32->bf16 %.1f us" % t(lambda: x.bfloat16()))
s = 8
dy = torch.randn(M, N, device=dev); dyb = dy.bfloat16()
print("fp32 splitK bmm %.1f us" % t(lambda: torch.bmm(dy.view(s, M // s, N).transpose(1, 2), x.view(s, M // s, K)).sum(0)))
print("bf16 splitK bmm %.1f us" % t(lambda: torch.bmm(dyb.view(s, M // s, N).transpose(1, 2), xb.view(s, M // s, K)).float().sum(0)))
try:
    print("bf16->f32 bmm   %.1f us" % t(lambda: torch.bmm(dyb.view(s, M // s, N).transpose(1, 2), xb.view(s, M // s, K), out_dtype=torch.float32).sum(0)))
except Exception as e:
    print("bmm out_dtype unsupported:", type(e).__name__, str(e)[:120])
print("bf16 full dW mm %.1f us" % t(lambda: dyb.t() @ xb))
