"""GPU: the bf16-operand GEMM family (csrc/gemm_bf16.h through the C-ABI: tmjx_bf16_shadow / tmjx_bgemm_nt / tmjx_bgemm_dw) against float64
torch on the shapes of BASELINE config 5 (40 960 rows; widths of the rodent-mc-intention nets, track_mjx/config/rodent-full-clips.yaml:50-57,
incl. the odd ones: 470 / 286 inputs, 120 / 76 / 1 outputs) and on ragged small shapes.  The operands are rounded to bf16 exactly as torch's
.to(bfloat16) does (round to nearest even), products of two bf16 are exact in fp32, so what remains is the fp32 accumulation error:
|err| <= c * eps32 * sum |a||b|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 2.0 ** -24


def _bf(x):
    return x.to(torch.bfloat16)


def test_shadows_are_torch_bf16_casts_with_zero_pads():
    from track_mjx_amd.agent.networks import Bf16Shadows, _dense
    torch.manual_seed(0)
    lins = [_dense(470, 1024).to(DEV), _dense(286, 512).to(DEV), _dense(256, 76).to(DEV), _dense(512, 1).to(DEV), _dense(7, 5).to(DEV)]
    # a row-padded parameter view, as the flat optimiser buffers hold the 470- / 286-wide layers
    wp = torch.zeros((1024, 472), device=DEV)
    wp[:, :470] = lins[0].weight.data
    lins[0].weight.data = wp[:, :470]
    sh = Bf16Shadows(lins)
    sh.refresh()
    torch.cuda.synchronize()
    for lin in lins:
        N, K = lin.weight.shape
        w, wt = sh.w[lin], sh.wt[lin]
        ref = _bf(lin.weight.detach())
        assert w.shape == (N, (K + 63) // 64 * 64) and wt.shape == (K, (N + 63) // 64 * 64)
        assert torch.equal(w[:, :K].view(torch.int16), ref.view(torch.int16))
        assert torch.equal(wt[:, :N].view(torch.int16), ref.t().contiguous().view(torch.int16))
        assert (w[:, K:].view(torch.int16) == 0).all() and (wt[:, N:].view(torch.int16) == 0).all()


@pytest.mark.parametrize("a_bf16", [False, True])
@pytest.mark.parametrize("M,N,K,lda", [(40960, 512, 512, 512), (20480, 1024, 470, 696), (40960, 256, 512, 512), (4096, 120, 512, 512), (3000, 76, 256, 256),
                                       (2048, 1, 256, 256), (1000, 512, 286, 288), (37, 5, 7, 8), (81, 130, 33, 40), (1, 1, 1, 8), (163, 600, 100, 104)])
def test_bgemm_nt(M, N, K, lda, a_bf16):
    from track_mjx_amd.agent.networks import Bf16Shadows, bgemm_nt, _dense
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    lin = _dense(K, N).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(torch.randn((N, K), generator=g, device=DEV) / K ** 0.5)
        lin.bias.copy_(torch.randn(N, generator=g, device=DEV))
    sh = Bf16Shadows([lin])
    sh.refresh()
    buf = torch.randn((M, lda), generator=g, device=DEV)
    if a_bf16:
        buf = buf.to(torch.bfloat16)
    x = buf[:, :K]
    y = bgemm_nt(x, sh.w[lin], N, K, lin.bias)
    torch.cuda.synchronize()
    a64, w64 = _bf(x).double(), _bf(lin.weight.detach()).double()
    ref = a64 @ w64.t() + lin.bias.detach().double()
    bound = (a64.abs() @ w64.abs().t()) * EPS * (K ** 0.5 + 4) * 2 + lin.bias.detach().abs().double() * EPS * 2 + 1e-30
    err = (y.double() - ref).abs()
    assert y.shape == (M, N) and (err <= bound).all(), float((err / bound).max())
    # input gradient form: dx = dy W through the transposed shadow
    dy = torch.randn((M, (N + 7) // 8 * 8), generator=g, device=DEV)[:, :N]        # (rows 16-byte aligned in either dtype)
    if a_bf16:
        dy = dy.to(torch.bfloat16)[:, :N] if dy.is_contiguous() else torch.randn((M, (N + 7) // 8 * 8), generator=g, device=DEV).to(torch.bfloat16)[:, :N]
    dx = bgemm_nt(dy, sh.wt[lin], K, N)
    d64 = _bf(dy).double()
    refx = d64 @ w64
    boundx = (d64.abs() @ w64.abs()) * EPS * (N ** 0.5 + 4) * 2 + 1e-30
    errx = (dx.double() - refx).abs()
    assert dx.shape == (M, K) and (errx <= boundx).all(), float((errx / boundx).max())


@pytest.mark.parametrize("y_bf16,x_bf16", [(False, False), (True, True), (False, True)])
@pytest.mark.parametrize("M,N,K,ldx,bias", [(40960, 512, 512, 512, True), (20480, 256, 470, 696, True), (4096, 120, 512, 512, True), (3000, 76, 256, 256, True),
                                            (2048, 1, 256, 256, True), (1000, 512, 286, 288, False), (333, 5, 7, 8, True), (81, 130, 33, 40, True), (1, 1, 1, 8, True)])
def test_bgemm_dw(M, N, K, ldx, bias, y_bf16, x_bf16):
    from track_mjx_amd.agent.networks import bgemm_dw
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + K)
    xb = torch.randn((M, ldx), generator=g, device=DEV)
    ldy = (N + 7) // 8 * 8
    yb = torch.randn((M, ldy), generator=g, device=DEV)
    if x_bf16:
        xb = xb.to(torch.bfloat16)
    if y_bf16:
        yb = yb.to(torch.bfloat16)
    x, dy = xb[:, :K], yb[:, :N]
    # destination with a leading dimension (row-padded flat-buffer view), pre-filled: the pads must stay untouched
    full = torch.full((N, (K + 3) // 4 * 4), 7.0, device=DEV)
    dw, db = bgemm_dw(dy, x, bias, out=full[:, :K])
    torch.cuda.synchronize()
    d64, x64 = _bf(dy).double(), _bf(x).double()
    ref = d64.t() @ x64
    bound = (d64.abs().t() @ x64.abs()) * EPS * (M ** 0.5 + 4) * 2 + 1e-30
    err = (dw.double() - ref).abs()
    assert (err <= bound).all(), float((err / bound).max())
    assert (full[:, K:] == 7.0).all()
    if bias:
        src = dy.double()          # the bias gradient sums the values as stored (fp32 values are NOT rounded to bf16 first)
        assert ((db.double() - src.sum(0)).abs() <= src.abs().sum(0) * EPS * (M ** 0.5 + 4) * 2 + 1e-30).all()
    else:
        assert db is None


def _ln_ref(z64, b64, g64, be64, eps):
    a = torch.nn.functional.silu(z64 + b64)
    return torch.nn.functional.layer_norm(a, (z64.shape[1],), g64, be64, eps), a


@pytest.mark.parametrize("a_bf16", [False, True])
@pytest.mark.parametrize("M,N,K,lda", [(40960, 512, 512, 512), (20480, 256, 470, 696), (1000, 128, 286, 288), (77, 512, 100, 104), (4096, 256, 256, 256)])
def test_bgemm_ln_forward_epilogue(M, N, K, lda, a_bf16):
    """tmjx_bgemm_ln_fwd: z against the float64 product of the bf16-rounded operands, y (bf16) and the row statistics against LayerNorm(silu(z + b))
    evaluated in float64 ON THE KERNEL'S OWN z (so what is tested is the epilogue, to one bf16 rounding)."""
    from track_mjx_amd.agent.networks import Bf16Shadows, _Block, bgemm_ln_fwd
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    blk = _Block(K, N).to(DEV)
    with torch.no_grad():
        blk.dense.bias.copy_(torch.randn(N, generator=g, device=DEV) * 0.3)
        blk.norm.weight.copy_(1 + 0.2 * torch.randn(N, generator=g, device=DEV))
        blk.norm.bias.copy_(0.2 * torch.randn(N, generator=g, device=DEV))
    sh = Bf16Shadows([blk.dense]); sh.refresh()
    buf = torch.randn((M, lda), generator=g, device=DEV)
    x = (buf.to(torch.bfloat16) if a_bf16 else buf)[:, :K]
    z, y, stats = bgemm_ln_fwd(x, sh.w[blk.dense], N, K, blk.dense.bias, blk.norm.weight, blk.norm.bias, 1e-6)
    torch.cuda.synchronize()
    a64, w64 = _bf(x).double(), _bf(blk.dense.weight.detach()).double()
    ref = a64 @ w64.t()
    bound = (a64.abs() @ w64.abs().t()) * EPS * (K ** 0.5 + 4) * 2 + 1e-30
    assert ((z.double() - ref).abs() <= bound).all()
    yr, a = _ln_ref(z.double(), blk.dense.bias.detach().double(), blk.norm.weight.detach().double(), blk.norm.bias.detach().double(), 1e-6)
    assert y.dtype == torch.bfloat16 and ((y.double() - yr).abs() <= 2.0 ** -8 * yr.abs() + 2e-5).all(), float((y.double() - yr).abs().max())
    mean, var = a.mean(1), a.var(1, unbiased=False)
    assert (stats[:, 0].double() - mean).abs().max() < 1e-5 and ((stats[:, 1].double() - (var + 1e-6).rsqrt()).abs() <= 1e-5 * (var + 1e-6).rsqrt()).all()


@pytest.mark.parametrize("dy_bf16", [False, True])
@pytest.mark.parametrize("M,N,K,ldy", [(40960, 512, 512, 512), (20480, 256, 120, 120), (3000, 512, 76, 80), (77, 128, 40, 40), (4096, 256, 256, 256)])
def test_bgemm_ln_backward_epilogue(M, N, K, ldy, dy_bf16):
    """tmjx_bgemm_ln_bwd: d loss / d z of a Dense -> SiLU -> LayerNorm block from the gradient dY of its consumer's output (the consumer's input-gradient
    GEMM + the block's LayerNorm / SiLU backward in the epilogue) and the (d gamma | d beta | d bias) column sums, against float64 autograd through
    LayerNorm(silu(z + b)) fed with the float64 product of the bf16-rounded operands."""
    from track_mjx_amd.agent.networks import Bf16Shadows, _dense, bgemm_ln_bwd
    g = torch.Generator(device=DEV).manual_seed(M + N + 7 * K)
    cons = _dense(N, K).to(DEV)                       # the consumer layer: N (the block's width) -> K
    sh = Bf16Shadows([cons]); sh.refresh()
    z = torch.randn((M, N), generator=g, device=DEV)
    b = torch.randn(N, generator=g, device=DEV) * 0.3
    gam = 1 + 0.2 * torch.randn(N, generator=g, device=DEV)
    a = torch.nn.functional.silu(z + b)
    stats = torch.stack([a.mean(1), (a.var(1, unbiased=False) + 1e-6).rsqrt()], 1).contiguous()
    buf = torch.randn((M, ldy), generator=g, device=DEV)
    dy = (buf.to(torch.bfloat16) if dy_bf16 else buf)[:, :K]
    dz, partial = bgemm_ln_bwd(dy, sh.wt[cons], N, K, z, b, gam, stats)
    torch.cuda.synchronize()
    gy = _bf(dy).double() @ _bf(cons.weight.detach()).double()                  # d loss / d y of the block, exact operands
    z64 = z.double().requires_grad_(True)
    b64, g64 = b.double().requires_grad_(True), gam.double().requires_grad_(True)
    be64 = torch.zeros(N, dtype=torch.float64, device=DEV, requires_grad=True)
    yr, _ = _ln_ref(z64, b64, g64, be64, 1e-6)
    rz, rb, rg, rbe = torch.autograd.grad(yr, [z64, b64, g64, be64], gy)
    scale = rz.abs().max()
    assert dz.dtype == torch.bfloat16 and ((dz.double() - rz).abs() <= 2.0 ** -8 * rz.abs() + 3e-5 * scale).all(), float(((dz.double() - rz).abs() / scale).max())
    sums = partial.double().sum(0)
    for got, ref in ((sums[0], rg), (sums[1], rbe), (sums[2], rb)):
        assert (got - ref).abs().max() <= 2e-4 * ref.abs().max() + 1e-6, float((got - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,N,K,lda,yf32", [(40960, 512, 696, 696, False), (4096, 256, 512, 512, True), (1000, 120, 100, 104, False), (77, 40, 33, 40, False)])
def test_bgemm_silu_forward_and_backward_epilogues(M, N, K, lda, yf32):
    """tmjx_bgemm_silu_fwd / _bwd (brax value MLP: Dense -> SiLU): z, y = silu(z + b) (bf16 or fp32), and dz = (dY Wc) silu'(z + b) with its column sums."""
    from track_mjx_amd.agent.networks import Bf16Shadows, _dense, bgemm_silu_bwd, bgemm_silu_fwd
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    lin, cons = _dense(K, N).to(DEV), _dense(N, 64).to(DEV)
    with torch.no_grad():
        lin.bias.copy_(torch.randn(N, generator=g, device=DEV) * 0.3)
    sh = Bf16Shadows([lin, cons]); sh.refresh()
    x = torch.randn((M, lda), generator=g, device=DEV)[:, :K]
    z, y = bgemm_silu_fwd(x, sh.w[lin], N, K, lin.bias, y_f32=yf32)
    torch.cuda.synchronize()
    a64, w64 = _bf(x).double(), _bf(lin.weight.detach()).double()
    ref = a64 @ w64.t()
    assert ((z.double() - ref).abs() <= (a64.abs() @ w64.abs().t()) * EPS * (K ** 0.5 + 4) * 2 + 1e-30).all()
    yr = torch.nn.functional.silu(z.double() + lin.bias.detach().double())
    tol = 1e-6 if yf32 else 2.0 ** -8
    assert y.dtype == (torch.float32 if yf32 else torch.bfloat16) and ((y.double() - yr).abs() <= tol * yr.abs() + 1e-6).all()
    dy = torch.randn((M, 64), generator=g, device=DEV)
    dz, partial = bgemm_silu_bwd(dy, sh.wt[cons], N, 64, z, lin.bias)
    torch.cuda.synchronize()
    gy = _bf(dy).double() @ _bf(cons.weight.detach()).double()
    v = z.double() + lin.bias.detach().double()
    sig = torch.sigmoid(v)
    rz = gy * (sig * (1 + v * (1 - sig)))
    assert ((dz.double() - rz).abs() <= 2.0 ** -8 * rz.abs() + 3e-5 * rz.abs().max()).all()
    assert (partial.double().sum(0) - rz.sum(0)).abs().max() <= 2e-3 * rz.sum(0).abs().max() + 1e-4
