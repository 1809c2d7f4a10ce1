"""GPU: the bf16-operand GEMM family (csrc/gemm_bf16.h through the C-ABI: tmjx_bf16_shadow / tmjx_bgemm_nt / tmjx_bgemm_dw) against float64
torch on the shapes of BASELINE config 5 (40 960 rows; widths of the rodent-mc-intention nets, track_mjx/config/rodent-full-clips.yaml:50-57,
incl. the odd ones: 470 / 286 inputs, 120 / 76 / 1 outputs) and on ragged small shapes.  The operands are rounded to bf16 exactly as torch's
.to(bfloat16) does (round to nearest even), products of two bf16 are exact in fp32, so what remains is the fp32 accumulation error:
|err| <= c * eps32 * sum |a||b|."""
import numpy as np
import pytest
import torch

from track_mjx_amd.agent import networks as nw

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


def test_grouped_weight_gradients_equal_the_layer_by_layer_ones():
    """deferred_weight_grads in bf16 GEMM-input mode: bgemm_dw calls with a destination view inside the block are recorded and computed by ONE
    tmjx_bgemm_dw_grouped launch at launch(); every result within the rounding bound of the float64 product (the slab split differs from the
    stand-alone call's, so the sums are taken in another order), pads of row-padded destinations untouched, a call WITHOUT a destination view
    inside the block is computed on the spot, 26 problems go as two groups."""
    from track_mjx_amd.agent.networks import bgemm_dw, deferred_weight_grads
    g = torch.Generator(device=DEV).manual_seed(77)
    shapes = [(8192, 512, 512, True, True, True), (8192, 1024, 470, True, False, True), (8192, 256, 512, False, True, True), (8192, 120, 512, True, True, False),
              (8192, 76, 256, True, True, True), (8192, 512, 286, True, False, True)] + [(4096, 128, 64 + 8 * i, True, i % 2 == 0, True) for i in range(20)]
    ops, outs = [], []
    with deferred_weight_grads() as d:
        for M, N, K, ybf, xbf, bias in shapes:
            ldx, ldy = (K + 7) // 8 * 8, (N + 7) // 8 * 8
            x = torch.randn((M, ldx), generator=g, device=DEV).to(torch.bfloat16 if xbf else torch.float32)[:, :K]
            dy = torch.randn((M, ldy), generator=g, device=DEV).to(torch.bfloat16 if ybf else torch.float32)[:, :N]
            full = torch.full((N, (K + 3) // 4 * 4), 7.0, device=DEV)
            dbv = torch.full((N,), 7.0, device=DEV) if bias else None
            dw, db = bgemm_dw(dy, x, bias, out=full[:, :K], out_bias=dbv)
            ops.append((dy, x, bias)); outs.append((full, dw, db))
        assert len(d.bproblems) == len(shapes)
        torch.cuda.synchronize()
        assert all(bool((f == 7.0).all()) for f, _, _ in outs), "a recorded problem was computed before launch()"
        # no destination view: computed on the spot
        dy, x, _ = ops[0]
        now, _ = bgemm_dw(dy, x, False)
        assert len(d.bproblems) == len(shapes)
        d.launch()
        assert not d.bproblems
    torch.cuda.synchronize()
    for (dy, x, bias), (full, dw, db) in zip(ops, outs):
        M, K = x.shape
        d64, x64 = _bf(dy).double(), _bf(x).double()
        ref = d64.t() @ x64
        bound = (d64.abs().t() @ x64.abs()) * EPS * (M ** 0.5 + 4) * 2 + 1e-30
        assert ((dw.double() - ref).abs() <= bound).all()
        assert (full[:, K:] == 7.0).all()
        if bias:
            src = dy.double()
            assert ((db.double() - src.sum(0)).abs() <= src.abs().sum(0) * EPS * (M ** 0.5 + 4) * 2 + 1e-30).all()
    d64, x64 = _bf(ops[0][0]).double(), _bf(ops[0][1]).double()
    assert ((now.double() - d64.t() @ x64).abs() <= (d64.abs().t() @ x64.abs()) * EPS * (8192 ** 0.5 + 4) * 2 + 1e-30).all()


def _ln_ref(z64, b64, g64, be64, eps):
    a = torch.nn.functional.silu(z64 + b64)
    return torch.nn.functional.layer_norm(a, (z64.shape[1],), g64, be64, eps), a


@pytest.mark.parametrize("a_bf16", [False, True])
@pytest.mark.parametrize("M,N,K,lda", [(40960, 512, 512, 512), (20480, 256, 470, 696), (1000, 128, 286, 288), (77, 512, 100, 104), (4096, 256, 256, 256)])
def test_bgemm_ln_forward_epilogue(M, N, K, lda, a_bf16):
    """tmjx_bgemm_ln_fwd: z (saved as bf16) against the float64 product of the bf16-rounded operands to one bf16 rounding, y (bf16) and the row
    statistics against LayerNorm(silu(. + b)) of that float64 product (they are computed from the fp32 accumulators, not from the rounded z)."""
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
    assert z.dtype == nw.bf16_z_dtype() and ((z.double() - ref).abs() <= bound + 2.0 ** -8 * ref.abs()).all()      # (bf16, or fp32 under -DTMJX_BF16_Z_F32)
    yr, a = _ln_ref(ref, blk.dense.bias.detach().double(), blk.norm.weight.detach().double(), blk.norm.bias.detach().double(), 1e-6)
    assert y.dtype == torch.bfloat16 and ((y.double() - yr).abs() <= 2.0 ** -8 * yr.abs() + 3e-5).all(), float((y.double() - yr).abs().max())
    mean, var = a.mean(1), a.var(1, unbiased=False)
    assert (stats[:, 0].double() - mean).abs().max() < 2e-5 and ((stats[:, 1].double() - (var + 1e-6).rsqrt()).abs() <= 2e-5 * (var + 1e-6).rsqrt()).all()


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
    z = torch.randn((M, N), generator=g, device=DEV).to(nw.bf16_z_dtype())        # the saved pre-activation as tmjx_bgemm_ln_fwd stores it (bf16; fp32 under -DTMJX_BF16_Z_F32)
    b = torch.randn(N, generator=g, device=DEV) * 0.3
    gam = 1 + 0.2 * torch.randn(N, generator=g, device=DEV)
    a = torch.nn.functional.silu(z.float() + b)
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
    assert z.dtype == nw.bf16_z_dtype() and ((z.double() - ref).abs() <= (a64.abs() @ w64.abs().t()) * EPS * (K ** 0.5 + 4) * 2 + 2.0 ** -8 * ref.abs() + 1e-30).all()
    yr = torch.nn.functional.silu(ref + lin.bias.detach().double())          # (y comes from the fp32 accumulators, not from the rounded z)
    tol = 2e-6 if yf32 else 2.0 ** -8
    assert y.dtype == (torch.float32 if yf32 else torch.bfloat16) and ((y.double() - yr).abs() <= tol * yr.abs() + 3e-6).all()
    dy = torch.randn((M, 64), generator=g, device=DEV)
    dz, partial = bgemm_silu_bwd(dy, sh.wt[cons], N, 64, z, lin.bias)
    torch.cuda.synchronize()
    gy = _bf(dy).double() @ _bf(cons.weight.detach()).double()
    v = z.double() + lin.bias.detach().double()
    sig = torch.sigmoid(v)
    rz = gy * (sig * (1 + v * (1 - sig)))
    assert ((dz.double() - rz).abs() <= 2.0 ** -8 * rz.abs() + 3e-5 * rz.abs().max()).all()
    assert (partial.double().sum(0) - rz.sum(0)).abs().max() <= 2e-3 * rz.sum(0).abs().max() + 1e-4


@pytest.mark.parametrize("kmajor,norm", [(False, False), (True, True), (False, True)])
@pytest.mark.parametrize("M,N,K", [(1368, 1024, 470), (2732, 512, 512), (1364, 256, 512), (1368, 120, 512), (1365, 76, 256), (100, 512, 286), (33, 40, 36)])
def test_linear_nolds_bf16(M, N, K, kmajor, norm):
    """tmjx_linear_nolds_bf16 (the acting policy's layer in bf16 GEMM-input mode, LDS-free): bf16-rounded activations x the bf16 shadow, fp32
    accumulate, against float64 torch on the SAME rounded operands; row-major and [k][row] (the env's observation buffer) activations, with and
    without the normaliser applied while the operand is loaded.  The learner's forward kernel on the same inputs must agree to fp32
    accumulation error: ONE set of numerics for acting and learning."""
    import ctypes as C
    from track_mjx_amd import hip as _hip
    from track_mjx_amd.agent.networks import Bf16Shadows, bgemm_nt, _dense
    if kmajor and M % 4:
        M = M // 4 * 4
    g = torch.Generator(device=DEV).manual_seed(M + N + K + int(kmajor))
    lin = _dense(K, N).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(torch.randn((N, K), generator=g, device=DEV) / K ** 0.5)
        lin.bias.copy_(torch.randn(N, generator=g, device=DEV))
    sh = Bf16Shadows([lin])
    sh.refresh()
    Kp = (K + 3) // 4 * 4
    if kmajor:
        buf = torch.randn((Kp + 8, M), generator=g, device=DEV)        # [k][row]: more rows than K, as the 696-row observation buffer behind the 470 reference rows
        x = buf[:Kp].t()                                              # logical [M][Kp]
        sa_row, sa_k = 1, M
    else:
        buf = torch.randn((M, Kp), generator=g, device=DEV)
        buf[:, K:] = 0                                                # (activation rows are zero padded to a multiple of 4)
        x = buf
        sa_row, sa_k = Kp, 1
    mean = torch.randn(Kp, generator=g, device=DEV) * 0.3 if norm else None
    istd = (torch.rand(Kp, generator=g, device=DEV) + 0.5) if norm else None
    if norm:
        mean[K:] = 0; istd[K:] = 0
    out = torch.empty((M, N), device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    w = sh.w[lin]
    _hip.check(_hip.lib().tmjx_linear_nolds_bf16(p(buf), sa_row, sa_k, p(w), w.stride(0), p(lin.bias), p(out), M, N, Kp, p(mean), p(istd),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tmjx_linear_nolds_bf16")
    torch.cuda.synchronize()
    xa = ((x - mean) * istd) if norm else x          # fp32, as the kernel forms it
    a64, w64 = _bf(xa[:, :K]).double(), _bf(lin.weight.detach()).double()
    ref = a64 @ w64.t() + lin.bias.detach().double()
    bound = (a64.abs() @ w64.abs().t()) * EPS * (K ** 0.5 + 4) * 2 + lin.bias.detach().abs().double() * EPS * 2 + 1e-30
    err = (out.double() - ref).abs()
    assert (err <= bound).all(), float((err / bound).max())
    # the learner's forward kernel (LDS tiles, v_cvt at the LDS write) on the same operands
    y = bgemm_nt(xa[:, :K].contiguous() if K % 4 == 0 else torch.nn.functional.pad(xa[:, :K], (0, Kp - K))[:, :K], w, N, K, lin.bias)
    torch.cuda.synchronize()
    assert ((y.double() - out.double()).abs() <= 2 * bound).all()


def test_bf16_mode_acting_and_learning_share_their_numerics():
    """BASELINE config 5's bf16 GEMM-input mode: the behaviour log-prob a roll-out stores (acting policy, LDS-free kernels) and the learner's
    first-pass log-prob of the same actions must come from ONE set of numerics, so that the PPO ratio starts at 1 (round-3 advisor finding:
    the acting path ran on the fp32 kernels and the ratio carried a bf16-sized bias).  (1) a pipelined roll-out (two env groups: the LDS-free
    acting path) in bf16 mode runs and stores finite log-probs; (2) the encoder stack through the acting path's kernels
    (tmjx_linear_nolds_bf16 + tmjx_silu_ln_fwd, layer by layer as PPOLearner._act_fused launches them) against the learner's bf16-mode forward
    of the same observations: agreement far inside the gap that the fp32 acting kernels leave."""
    import ctypes as C
    from tests.common import make_env_and_oracle
    from track_mjx_amd import hip as _hip
    from track_mjx_amd.agent import ppo
    from track_mjx_amd.agent.networks import gemm_inputs
    envs = [make_env_and_oracle(num_envs=128, n_clips=4, wrappers=True, seed=k)[0] for k in range(2)]
    L = ppo.PPOLearner(envs, encoder_layers=(256, 128), decoder_layers=(128, 128), critic_layers=(128, 128), latents=60, unroll_length=4,
                       batch_size=256, num_minibatches=1, num_updates_per_batch=1, seed=5, matmul_dtype=torch.bfloat16)
    assert L.shadows is not None and L.lds_free
    for k, e in enumerate(envs):
        L.states[k] = e.reset(torch.Generator().manual_seed(20 + k))
    L.collect()
    torch.cuda.synchronize()
    assert torch.isfinite(L.buf["log_prob"]).all() and torch.isfinite(L.buf["raw_action"]).all()
    obs = L.buf["observation"][0]
    ref_w = L.policy.reference_obs_size
    h = torch.nn.functional.pad(L.normalizer.normalize(obs)[:, :ref_w], (0, (-ref_w) % 4)).contiguous()      # rows padded to a multiple of 4 floats (zeros)
    x = h[:, :ref_w]                                                                                          # (16-byte aligned rows, as the learner's gather makes them)
    with torch.no_grad(), gemm_inputs(L.matmul_dtype, L.shadows):
        fc2_learn = L.policy.fc2(L.policy.encoder(x))
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = x.shape[0]
    lib = _hip.lib()
    for blk in L.policy.encoder:
        w = L.shadows.w[blk.dense]
        z = torch.empty((n, blk.dense.out_features), device=DEV)
        _hip.check(lib.tmjx_linear_nolds_bf16(p(h), h.shape[1], 1, p(w), w.stride(0), None, p(z), n, blk.dense.out_features, h.shape[1], None, None, stream), "nolds_bf16")
        y = torch.empty_like(z); stats = torch.empty((n, 2), device=DEV)
        _hip.check(lib.tmjx_silu_ln_fwd(p(z), p(blk.dense.bias), p(blk.norm.weight), p(blk.norm.bias), p(y), p(stats), n, blk.dense.out_features, float(blk.norm.eps), stream), "silu_ln")
        h = y
    w = L.shadows.w[L.policy.fc2]
    fc2_act = torch.empty((n, L.policy.fc2.out_features), device=DEV)
    _hip.check(lib.tmjx_linear_nolds_bf16(p(h), h.shape[1], 1, p(w), w.stride(0), p(L.policy.fc2.bias), p(fc2_act), n, L.policy.fc2.out_features, h.shape[1], None, None, stream), "nolds_bf16")
    with torch.no_grad():
        fc2_f32 = L.policy.fc2(L.policy.encoder(x))           # the fp32 kernels: what the acting path ran on before
    torch.cuda.synchronize()
    scale = fc2_learn.abs().max().item()
    d_bf, d_f32 = (fc2_act - fc2_learn).abs().max().item() / scale, (fc2_f32 - fc2_learn).abs().max().item() / scale
    print(f"\nbf16 mode, encoder output against the learner's forward pass (relative): bf16 acting kernels {d_bf:.2e}, fp32 acting kernels {d_f32:.2e}")
    assert d_bf <= 0.25 * d_f32 and d_bf < 2e-3


@pytest.mark.parametrize("H", [64, 256, 512, 1024])
def test_silu_ln_bf16_results_are_the_rounded_fp32_results(H):
    """tmjx_silu_ln_fwd_bf16 / _bwd_bf16 (the row kernels of a Dense -> SiLU -> LayerNorm block that is not one GEMM tile wide, with a bf16
    result for the bf16-operand GEMMs that consume it): bit for bit torch's .to(bfloat16) (round to nearest even) of what tmjx_silu_ln_fwd /
    _bwd write as fp32 — the bits those GEMMs would have made when they stage the fp32 rows — with the same fp32 statistics and column sums.
    Ragged row count (1000: the last 32-row block of the backward kernel is partial), output leading dimension wider than H."""
    import ctypes as C
    from track_mjx_amd import hip as _hip
    lib = _hip.lib()
    g = torch.Generator(device=DEV).manual_seed(H)
    rows, ld = 1000, H + 8
    z, dy = torch.randn(rows, H, generator=g, device=DEV), torch.randn(rows, H, generator=g, device=DEV)
    bias, gamma, beta = 0.3 * torch.randn(H, generator=g, device=DEV), 1 + 0.2 * torch.randn(H, generator=g, device=DEV), 0.1 * torch.randn(H, generator=g, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y, st = torch.empty_like(z), torch.empty(rows, 2, device=DEV)
    y16, st16 = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=DEV), torch.empty(rows, 2, device=DEV)
    _hip.check(lib.tmjx_silu_ln_fwd(p(z), p(bias), p(gamma), p(beta), p(y), p(st), rows, H, 1e-6, stream), "fwd")
    _hip.check(lib.tmjx_silu_ln_fwd_bf16(p(z), p(bias), p(gamma), p(beta), p(y16), ld, p(st16), rows, H, 1e-6, stream), "fwd16")
    assert torch.equal(y16[:, :H], y.to(torch.bfloat16)) and torch.equal(st, st16) and bool((y16[:, H:] == 7.0).all())
    y_ref = torch.nn.functional.layer_norm(torch.nn.functional.silu(z.double() + bias.double()), (H,), gamma.double(), beta.double(), 1e-6)
    assert float((y.double() - y_ref).abs().max()) < 2e-5
    npart = int(lib.tmjx_silu_ln_partial_floats(rows, H))
    dz, g3, part = torch.empty_like(z), torch.empty(3, H, device=DEV), torch.empty(npart, device=DEV)
    dz16, g316, part16 = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device=DEV), torch.empty(3, H, device=DEV), torch.empty(npart, device=DEV)
    _hip.check(lib.tmjx_silu_ln_bwd(p(dy), p(z), p(bias), p(gamma), p(st), p(dz), p(g3), p(part), rows, H, stream), "bwd")
    _hip.check(lib.tmjx_silu_ln_bwd_bf16(p(dy), p(z), p(bias), p(gamma), p(st), p(dz16), ld, p(g316), p(part16), rows, H, stream), "bwd16")
    assert torch.equal(dz16[:, :H], dz.to(torch.bfloat16)) and torch.equal(g3, g316) and bool((dz16[:, H:] == 7.0).all())
    # error paths: a misaligned / too narrow bf16 destination is refused
    assert lib.tmjx_silu_ln_fwd_bf16(p(z), p(bias), p(gamma), p(beta), p(y16), H - 4, p(st16), rows, H, 1e-6, stream) != 0
    assert lib.tmjx_silu_ln_bwd_bf16(p(dy), p(z), p(bias), p(gamma), p(st), C.c_void_p(dz16.data_ptr() + 2), ld, p(g316), p(part16), rows, H, stream) != 0


def test_minibatch_begin_bf16_twin_and_the_chain_that_stages_it():
    """bf16 GEMM-input mode feeds its first layers from a bf16 TWIN of the minibatch's normalised observations, written by the gather launch
    (tmjx_minibatch_begin_bf16) — bit-identical by construction: (1) the twin is torch's .to(bfloat16) of the fp32 rows, zero beyond the
    observation width; (2) one SGD step's gradient buffer with the twin (first layers by LDS-DMA over whole K tiles, the 1024-wide encoder block's
    row kernels with bf16 results) equals the step without it (TMJX_NO_BF16_TWIN=1: fp32 rows converted while they are staged) BIT FOR BIT."""
    import os
    from tests.common import make_env_and_oracle
    from track_mjx_amd.agent import ppo
    env = make_env_and_oracle(num_envs=256, n_clips=4, wrappers=True, seed=3)[0]
    L = ppo.PPOLearner(env, encoder_layers=(1024, 512), decoder_layers=(512, 256), critic_layers=(512, 256), latents=60, unroll_length=4,
                       batch_size=256, num_minibatches=1, num_updates_per_batch=1, seed=11, matmul_dtype=torch.bfloat16, use_graph=False)
    assert L.shadows is not None and L._obs16 is not None and L._obs16.shape == (4 * 256, 704) and L._self_advancing()
    L.states[0] = env.reset(torch.Generator().manual_seed(1))
    L.collect()
    L.normalizer.update(L.buf["observation"])
    L._perm_static.copy_(torch.randperm(L._perm_static.numel(), generator=torch.Generator().manual_seed(2)).to(DEV))
    state0 = L._mb_state.clone()
    grads = []
    for no_twin in (False, True):
        L._mb_state.copy_(state0)
        L._acc8.zero_()
        if no_twin:
            os.environ["TMJX_NO_BF16_TWIN"] = "1"
        try:
            L._minibatch_grads(None, 0.05)
        finally:
            os.environ.pop("TMJX_NO_BF16_TWIN", None)
        torch.cuda.synchronize()
        grads.append((L.grads.flat.clone(), L._acc8.clone()))
    assert torch.isfinite(grads[0][0]).all() and float(grads[0][0].abs().max()) > 0
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    # (1): the twin against the fp32 rows of the same launch
    L._mb_state.copy_(state0)
    data = L._mb_data(None)
    on, tw = data["observation_normalized"], data["observation_normalized_bf16"]
    W = on.shape[-1]
    assert torch.equal(tw[:, :W], on.reshape(-1, W).to(torch.bfloat16)) and bool((tw[:, W:] == 0).all())


@pytest.mark.parametrize("M,N", [(1000, 256), (163, 120), (4096, 1024)])
def test_bf_silu_bwd_rank1_is_the_outer_product_through_the_plain_kernel(M, N):
    """tmjx_bf_silu_bwd_rank1 (the value head's input gradient dy1 x w1 formed inside the last hidden layer's SiLU backward) against
    tmjx_bf_silu_bwd on the materialised outer product: d loss / d z and the per-tile column sums bit for bit (one fp32 product per element either
    way), and both against float64 torch."""
    import ctypes as C
    from track_mjx_amd import hip as _hip
    lib = _hip.lib()
    g = torch.Generator(device=DEV).manual_seed(M + N)
    dy1, w1 = torch.randn(M, generator=g, device=DEV), torch.randn(N, generator=g, device=DEV) * 0.1
    z, bias = torch.randn(M, N, generator=g, device=DEV).to(nw.bf16_z_dtype()), 0.3 * torch.randn(N, generator=g, device=DEV)
    dy = (dy1[:, None] * w1[None, :]).contiguous()
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    npart = int(lib.tmjx_bgemm_partial_floats(M, N, 1))
    ld = (N + 7) // 8 * 8
    out = []
    for rank1 in (False, True):
        dz, part = torch.zeros((M, ld), dtype=torch.bfloat16, device=DEV), torch.empty(npart, device=DEV)
        if rank1:
            _hip.check(lib.tmjx_bf_silu_bwd_rank1(p(dy1), p(w1), p(z), N, p(bias), p(dz), ld, p(part), M, N, stream), "rank1")
        else:
            _hip.check(lib.tmjx_bf_silu_bwd(p(dy), N, p(z), N, p(bias), p(dz), ld, p(part), M, N, stream), "plain")
        out.append((dz, part))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    v = z.double() + bias.double()
    sig = torch.sigmoid(v)
    ref = dy.double() * (sig * (1 + v * (1 - sig)))
    assert torch.equal(out[1][0][:, :N], ref.float().to(torch.bfloat16)) or float((out[1][0][:, :N].double() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max())
    tiles = (M + 79) // 80
    ref_part = torch.stack([ref[80 * t:80 * (t + 1)].sum(0) for t in range(tiles)])
    assert float((out[1][1].view(tiles, N).double() - ref_part).abs().max()) <= 1e-5 * float(ref_part.abs().max())
    assert lib.tmjx_bf_silu_bwd_rank1(p(dy1), p(w1), p(z), N, p(bias), p(out[1][0]), ld, p(out[1][1]), M, N + 2, stream) != 0     # N not a multiple of 4


def test_value_net_head_inside_the_bf16_chain_changes_no_bit():
    """bf16 GEMM-input mode: the value MLP's 1-wide head runs inside the chain (its input gradient is never materialised: tmjx_bf_silu_bwd_rank1);
    output and every parameter gradient equal the stand-alone head layer's (TMJX_NO_CHAIN_HEAD=1: tmjx_gemm_nn + tmjx_bf_silu_bwd) bit for bit."""
    import os
    from track_mjx_amd.agent.networks import Bf16Shadows, ValueNet, gemm_inputs
    torch.manual_seed(3)
    g = torch.Generator(device=DEV).manual_seed(4)
    obs, up = torch.randn(5, 400, 696, generator=g, device=DEV), torch.randn(5, 400, generator=g, device=DEV)
    res = []
    for no_head in (False, True):
        torch.manual_seed(3)
        net = ValueNet(696, (512, 256)).to(DEV)
        lins = [m for m in net.modules() if isinstance(m, torch.nn.Linear) and m.out_features % 4 == 0]
        sh = Bf16Shadows(lins, need_t=lins[1:])
        sh.refresh()
        if no_head:
            os.environ["TMJX_NO_CHAIN_HEAD"] = "1"
        try:
            with gemm_inputs(torch.bfloat16, sh):
                y = net(obs)
                grads = torch.autograd.grad((y * up).sum(), list(net.parameters()))
        finally:
            os.environ.pop("TMJX_NO_CHAIN_HEAD", None)
        assert (net._chain[-1].kind == "head") == (not no_head)
        res.append([y.detach()] + [t.clone() for t in grads])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bf16_mode_small_minibatch_takes_the_padded_copy():
    """bf16 GEMM-input mode with a minibatch below the fused latent kernel's threshold (< 1024 rows): the decoder's input is a torch.cat of 286
    columns, whose rows do not start on 16-byte boundaries — the chain stages a copy with padded rows (it used to hand the rows to tmjx_bgemm_ln_fwd
    as they were: TMJX_EINVAL).  One update through the captured step: finite, and the gradient agrees with the fp32 learner's."""
    from tests.common import make_env_and_oracle
    from track_mjx_amd.agent import ppo
    env = make_env_and_oracle(num_envs=128, n_clips=4, wrappers=True)[0]
    kw = dict(encoder_layers=(256, 128), decoder_layers=(128, 128), critic_layers=(128, 128), latents=60, unroll_length=5, batch_size=128,
              num_minibatches=4, num_updates_per_batch=1, seed=9)
    L32 = ppo.PPOLearner(env, use_graph=False, **kw)
    L16 = ppo.PPOLearner(env, matmul_dtype=torch.bfloat16, **kw)
    L32.states[0] = env.reset(torch.Generator().manual_seed(1))
    L32.collect()
    for k in L32.buf:
        L16.buf[k].copy_(L32.buf[k])
    for n in (L32, L16):
        n.normalizer.update(n.buf["observation"])
    idx = torch.arange(L32.local_batch, device=env.device)
    torch.manual_seed(0)
    L32._minibatch_grads(idx, 0.1)
    torch.manual_seed(0)
    L16._minibatch_grads(idx, 0.1)
    g32, g16 = L32.grads.flat, L16.grads.flat
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    assert torch.isfinite(g16).all() and cos > 0.98, cos
    draw0 = int(L16._mb_state[0])
    out = L16.update(0)
    torch.cuda.synchronize()
    assert L16._graph is not None, "hipGraph capture of the small-minibatch bf16 step failed"
    assert all(bool(torch.isfinite(v).all()) for v in out.values())
    assert int(L16._mb_state[0]) == draw0 + 4          # one draw per minibatch step: the capture's warm-up runs leave the draw counter where it was
