"""GPU: the learner's MFMA GEMM kernels (csrc/gemm_kernels.h through the C-ABI: tmjx_gemm_nt / _nn / _dw) against float64 torch
on the shapes the PPO step uses (rows = 20 x 1024; widths of the 2x256 and rodent-mc-intention nets incl. the odd ones: 470 / 286
inputs with a leading dimension of 696, 120 / 76 / 1 outputs) and on ragged small shapes.  fp32 MFMA = a k-ordered fmaf chain, so
the error bound is the fp32 dot-product one: |err| <= c * eps * sum |a||b|."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 2.0 ** -24


def _bound(a64, b64_t, K):
    return (a64.abs() @ b64_t.abs()) * EPS * (K ** 0.5 + 4) * 2 + 1e-30


@pytest.mark.parametrize("M,N,K,lda", [(20480, 256, 470, 696), (20480, 256, 256, 256), (20480, 120, 256, 256), (20480, 76, 256, 256), (20480, 1, 256, 256),
                                       (20480, 512, 1024, 1024), (1000, 256, 286, 288), (37, 5, 7, 7), (81, 130, 33, 36), (1, 1, 1, 1),
                                       # 5 120 rows = one rank's share of BASELINE configs[2] (batch_size 2048 / 8 GPUs x unroll_length 20): the 32-row tile (gemm_mt)
                                       (5120, 256, 470, 696), (5120, 256, 256, 256), (5120, 120, 256, 256), (5120, 76, 256, 256), (5117, 256, 286, 288), (5120, 1, 256, 256)])
def test_gemm_nt_forward(M, N, K, lda):
    from track_mjx_amd.agent.networks import gemm_nt
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    buf = torch.randn((M, lda), generator=g, device=DEV)
    x = buf[:, :K]
    w = torch.randn((N, K), generator=g, device=DEV) / K ** 0.5
    b = torch.randn(N, generator=g, device=DEV)
    y = gemm_nt(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    assert y.shape == (M, N)
    assert ((y.double() - ref).abs() <= _bound(x.double(), w.double().t(), K) + EPS * b.abs().double()).all(), float((y.double() - ref).abs().max())
    y0 = gemm_nt(x, w, None)
    assert torch.equal(y0 + b, y) or (y0.double() + b.double() - ref).abs().max() < 1e-4


@pytest.mark.parametrize("M,N,K,cols", [(20480, 256, 256, None), (20480, 120, 256, None), (20480, 256, 286, 60), (20480, 1, 256, None), (20480, 76, 256, None),
                                        (20480, 1024, 512, None), (53, 9, 21, None), (53, 9, 21, 5),
                                        (5120, 256, 256, None), (5120, 120, 256, None), (5120, 256, 286, 60), (5120, 76, 256, None), (5117, 1, 256, None)])
def test_gemm_nn_input_gradient(M, N, K, cols):
    from track_mjx_amd.agent.networks import gemm_nn
    g = torch.Generator(device=DEV).manual_seed(7 * M + N + K)
    dy = torch.randn((M, N), generator=g, device=DEV)
    w = torch.randn((N, K), generator=g, device=DEV) / N ** 0.5
    dx = gemm_nn(dy, w, cols)
    c = K if cols is None else cols
    ref = dy.double() @ w.double()
    assert dx.shape == (M, K)
    assert ((dx[:, :c].double() - ref[:, :c]).abs() <= _bound(dy.double(), w.double(), N)[:, :c]).all()


@pytest.mark.parametrize("M,N,K,ldx,bias", [(20480, 256, 470, 696, True), (20480, 256, 256, 256, True), (20480, 120, 256, 256, True), (20480, 1, 256, 256, True),
                                            (20480, 256, 696, 696, True), (20480, 76, 256, 256, False), (40960, 512, 1024, 1024, True), (333, 76, 286, 288, True),
                                            (31, 3, 5, 5, True), (20480, 256, 286, 288, True),
                                            (5120, 256, 470, 696, True), (5120, 256, 256, 256, True), (5120, 76, 256, 256, False), (5120, 256, 696, 696, True), (5117, 120, 256, 256, True)])
def test_gemm_dw_weight_and_bias_gradient(M, N, K, ldx, bias):
    from track_mjx_amd.agent.networks import gemm_dw
    g = torch.Generator(device=DEV).manual_seed(3 * M + N + K)
    xb = torch.randn((M, ldx), generator=g, device=DEV)
    x = xb[:, :K]
    dy = torch.randn((M, N), generator=g, device=DEV)
    dw, db = gemm_dw(dy, x, bias)
    ref = dy.double().t() @ x.double()
    bound = (dy.double().abs().t() @ x.double().abs()) * EPS * (M ** 0.5 + 4) * 2
    assert dw.shape == (N, K) and ((dw.double() - ref).abs() <= bound).all(), float(((dw.double() - ref).abs() / bound).max())
    if bias:
        refb = dy.double().sum(0)
        assert ((db.double() - refb).abs() <= dy.double().abs().sum(0) * EPS * (M ** 0.5 + 4) * 2).all()
    else:
        assert db is None


def test_dense_layers_use_no_library_gemm_and_match_torch_gradients():
    """A Dense -> SiLU -> LayerNorm block and a bias layer under autograd: outputs and all gradients against plain torch in float64."""
    from track_mjx_amd.agent.networks import _Block, _dense
    torch.manual_seed(0)
    blk, lin = _Block(470, 256).to(DEV), _dense(256, 120).to(DEV)
    obs = torch.randn((20, 1024, 696), device=DEV)
    x = obs[..., :470]
    y = lin(blk(x))
    gy = torch.randn_like(y)
    grads = torch.autograd.grad(y, [blk.dense.weight, blk.dense.bias, blk.norm.weight, blk.norm.bias, lin.weight, lin.bias], gy)
    xd = x.double()
    W0, b0, gam, bet, W1, b1 = [p.detach().double().requires_grad_() for p in (blk.dense.weight, blk.dense.bias, blk.norm.weight, blk.norm.bias, lin.weight, lin.bias)]
    z = torch.nn.functional.silu(xd @ W0.t() + b0)
    h = torch.nn.functional.layer_norm(z, (256,), gam, bet, 1e-6)
    yr = h @ W1.t() + b1
    gr = torch.autograd.grad(yr, [W0, b0, gam, bet, W1, b1], gy.double())
    assert (y.double() - yr).abs().max() < 1e-4
    for a, b in zip(grads, gr):
        assert (a.double() - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, float((a.double() - b).abs().max() / b.abs().max())


def test_grouped_weight_gradients_match_individual_problems():
    """tmjx_gemm_dw_grouped: several layers' (dW, db) in one launch + one reduction, results written with a leading dimension (row-padded
    flat-buffer views), against float64 — through the same recording context the learner uses."""
    from track_mjx_amd.agent.networks import deferred_weight_grads
    g = torch.Generator(device=DEV).manual_seed(0)
    M = 20480
    shapes = [(256, 470, 696, True), (256, 256, 256, True), (120, 256, 256, True), (76, 256, 256, False), (256, 286, 288, True)]
    probs = []
    with deferred_weight_grads() as d:
        for N, K, ldx, bias in shapes:
            xb = torch.randn((M, ldx), generator=g, device=DEV)
            x, dy = xb[:, :K], torch.randn((M, N), generator=g, device=DEV)
            w = torch.nn.Parameter(torch.zeros((N, (K + 3) // 4 * 4), device=DEV)[:, :K])
            w.grad = torch.full((N, (K + 3) // 4 * 4), 7.0, device=DEV)[:, :K]         # a strided destination, pre-filled: pads must stay untouched
            b = torch.nn.Parameter(torch.zeros(N, device=DEV)) if bias else None
            if b is not None:
                b.grad = torch.zeros(N, device=DEV)
            assert d.try_add(dy, x, w, b) is not None
            probs.append((dy, x, w, b))
        # unaligned operands are refused (the caller computes them individually)
        assert d.try_add(torch.randn((M, 1), device=DEV), probs[1][1], torch.nn.Parameter(torch.zeros((1, 256), device=DEV)), None) is None
    d.launch()
    torch.cuda.synchronize()
    for dy, x, w, b in probs:
        ref = dy.double().t() @ x.double()
        bound = (dy.double().abs().t() @ x.double().abs()) * EPS * (M ** 0.5 + 4) * 2
        assert ((w.grad.double() - ref).abs() <= bound).all()
        if w.grad.stride(0) != w.shape[1]:
            full = torch.as_strided(w.grad, (w.shape[0], w.grad.stride(0)), (w.grad.stride(0), 1))
            assert (full[:, w.shape[1]:] == 7.0).all()
        if b is not None:
            assert ((b.grad.double() - dy.double().sum(0)).abs() <= dy.double().abs().sum(0) * EPS * (M ** 0.5 + 4) * 2).all()


@pytest.mark.parametrize("M,N,K,lda", [(20480, 256, 470, 696), (20480, 256, 286, 288), (20480, 128, 256, 256), (1000, 64, 100, 100), (77, 256, 36, 36),
                                       (5120, 256, 470, 696), (5120, 256, 286, 288), (5120, 128, 256, 256), (5117, 64, 256, 256)])
def test_fused_dense_silu_layernorm_block(M, N, K, lda, monkeypatch):
    """tmjx_gemm_nt_silu_ln (the SiLU + LayerNorm epilogue on the GEMM tile) against the two-launch path (tmjx_gemm_nt + tmjx_silu_ln_fwd)
    and against float64 torch, forward and every gradient (reference block: intention_network.py:32-40 Dense -> silu -> LayerNorm)."""
    from track_mjx_amd.agent import networks as nw
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    buf = torch.randn((M, lda), generator=g, device=DEV)
    blk = nw._Block(K, N).to(DEV)
    with torch.no_grad():
        wpad = torch.zeros((N, (K + 3) // 4 * 4), device=DEV)          # rows 16-byte aligned, as in the learner's flat parameter buffer
        wpad[:, :K] = blk.dense.weight
        blk.dense.weight.data = wpad[:, :K]
        blk.dense.bias.copy_(torch.randn(N, generator=g, device=DEV) * 0.3)
        blk.norm.weight.copy_(1 + 0.2 * torch.randn(N, generator=g, device=DEV))
        blk.norm.bias.copy_(0.2 * torch.randn(N, generator=g, device=DEV))
    cot = torch.randn((M, N), generator=g, device=DEV)

    def run(fused):
        if fused:
            monkeypatch.delenv("TMJX_NO_FUSED_BLOCK", raising=False)
        else:
            monkeypatch.setenv("TMJX_NO_FUSED_BLOCK", "1")
        x = buf[:, :K].detach().requires_grad_(True)
        assert nw._block_fusable(x, blk.dense.weight) == fused
        y = blk(x)
        grads = torch.autograd.grad(y, [x, blk.dense.weight, blk.dense.bias, blk.norm.weight, blk.norm.bias], cot)
        return [y.detach()] + [t.detach() for t in grads]

    got, two = run(True), run(False)
    x64 = buf[:, :K].double().detach().requires_grad_(True)
    p64 = [p.detach().double().requires_grad_(True) for p in (blk.dense.weight, blk.dense.bias, blk.norm.weight, blk.norm.bias)]
    y64 = torch.nn.functional.layer_norm(torch.nn.functional.silu(x64 @ p64[0].t() + p64[1]), (N,), p64[2], p64[3], 1e-6)
    ref = [y64.detach()] + list(torch.autograd.grad(y64, [x64] + p64, cot.double()))
    names = ["y", "dx", "dW", "db", "dgamma", "dbeta"]
    for name, a, b, r in zip(names, got, two, ref):
        scale = r.abs().max().item() + 1e-30
        e_f, e_t = (a.double() - r).abs().max().item() / scale, (b.double() - r).abs().max().item() / scale
        print(f"{name}: fused {e_f:.2e}  two-launch {e_t:.2e} (rel. to max |ref|)")
        assert e_f <= max(3 * e_t, 2e-6), name


@pytest.mark.parametrize("M,widths", [(20480, (470, 256, 256, 120)), (1000, (286, 256, 256, 76)), (163, (64, 256, 256, 256, 40)),
                                      (5120, (470, 256, 256, 120)), (5117, (286, 256, 256, 76))])
def test_layernorm_backward_fused_into_the_next_input_gradient(M, widths, monkeypatch):
    """tmjx_gemm_nn_ln_bwd: the LayerNorm + SiLU backward of a 256-wide block in the epilogue of its consumer's input-gradient GEMM (inside
    `ln_bwd_links()`), against the unfused chain (tmjx_gemm_nn + tmjx_silu_ln_bwd) and float64 torch: every parameter gradient and dx."""
    from track_mjx_amd.agent import networks as nw
    g = torch.Generator(device=DEV).manual_seed(M + sum(widths))
    blocks = []
    for i, o in zip(widths[:-2], widths[1:-1]):
        blk = nw._Block(i, o).to(DEV)
        with torch.no_grad():
            wpad = torch.zeros((o, (i + 3) // 4 * 4), device=DEV)
            wpad[:, :i] = blk.dense.weight
            blk.dense.weight.data = wpad[:, :i]
            blk.dense.bias.copy_(torch.randn(o, generator=g, device=DEV) * 0.3)
            blk.norm.weight.copy_(1 + 0.2 * torch.randn(o, generator=g, device=DEV))
            blk.norm.bias.copy_(0.2 * torch.randn(o, generator=g, device=DEV))
        blocks.append(blk)
    head = nw._dense(widths[-2], widths[-1]).to(DEV)
    with torch.no_grad():
        head.bias.copy_(torch.randn(widths[-1], generator=g, device=DEV) * 0.1)
    params = [p for b in blocks for p in (b.dense.weight, b.dense.bias, b.norm.weight, b.norm.bias)] + [head.weight, head.bias]
    xbuf = torch.randn((M, (widths[0] + 3) // 4 * 4), generator=g, device=DEV)
    cot = torch.randn((M, widths[-1]), generator=g, device=DEV)

    def run(fused):
        if fused:
            monkeypatch.delenv("TMJX_NO_LN_BWD_FUSION", raising=False)
        else:
            monkeypatch.setenv("TMJX_NO_LN_BWD_FUSION", "1")
        x = xbuf[:, :widths[0]].detach().requires_grad_(True)
        with nw.ln_bwd_links():
            h = x
            for b in blocks:
                h = b(h)
            y = head(h)
        return [t.detach() for t in torch.autograd.grad(y, [x] + params, cot)]

    calls = []
    orig = nw._dx_through_block
    monkeypatch.setattr(nw, "_dx_through_block", lambda *a: (calls.append(1), orig(*a))[1])
    got = run(True)
    assert len(calls) == len(blocks), "every 256-wide block's backward must have gone through its consumer's epilogue"
    two = run(False)
    assert len(calls) == len(blocks)
    x64 = xbuf[:, :widths[0]].double().detach().requires_grad_(True)
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    h = x64
    for k in range(len(blocks)):
        w, b, gm, be = p64[4 * k:4 * k + 4]
        h = torch.nn.functional.layer_norm(torch.nn.functional.silu(h @ w.t() + b), (w.shape[0],), gm, be, 1e-6)
    y64 = h @ p64[-2].t() + p64[-1]
    ref = torch.autograd.grad(y64, [x64] + p64, cot.double())
    worst = 0.0
    for i, (a, b, r) in enumerate(zip(got, two, ref)):
        scale = r.abs().max().item() + 1e-30
        e_f, e_t = (a.double() - r).abs().max().item() / scale, (b.double() - r).abs().max().item() / scale
        worst = max(worst, e_f)
        assert e_f <= max(3 * e_t, 3e-6), (i, e_f, e_t)
    print(f"M={M} widths={widths}: worst relative error of the fused chain {worst:.2e}")


def test_weight_used_twice_under_deferred_weight_gradients():
    """A layer applied to two inputs inside deferred_weight_grads(): autograd ADDS the two uses' gradients, so the view handed out for the
    first use must hold that use's gradient by then (the recorded problem is launched on the spot, the second use gets a fresh tensor)."""
    from track_mjx_amd.agent.networks import _dense, deferred_weight_grads
    torch.manual_seed(0)
    lin = _dense(256, 128).to(DEV)
    lin.weight.grad, lin.bias.grad = torch.full_like(lin.weight, 3.0), torch.full_like(lin.bias, 3.0)     # stale contents of the gradient views
    x1, x2 = torch.randn((2048, 256), device=DEV), torch.randn((2048, 256), device=DEV)
    g1, g2 = torch.randn((2048, 128), device=DEV), torch.randn((2048, 128), device=DEV)
    with deferred_weight_grads() as d:
        gw, gb = torch.autograd.grad([lin(x1), lin(x2)], [lin.weight, lin.bias], [g1, g2])
    d.launch()
    torch.cuda.synchronize()
    rw = g1.double().t() @ x1.double() + g2.double().t() @ x2.double()
    rb = g1.double().sum(0) + g2.double().sum(0)
    assert (gw.double() - rw).abs().max() <= 1e-5 * rw.abs().max(), float((gw.double() - rw).abs().max())
    assert (gb.double() - rb).abs().max() <= 1e-5 * rb.abs().max()


def test_second_backward_through_a_retained_fused_chain():
    """retain_graph=True and a second backward through Dense->SiLU->LayerNorm blocks whose LayerNorm backward was fused into the consumer's
    input-gradient GEMM the first time: the second pass (unfused: the link's context is gone) must give the same gradients."""
    from track_mjx_amd.agent import networks as nw
    torch.manual_seed(1)
    blocks = [nw._Block(64, 256).to(DEV), nw._Block(256, 256).to(DEV)]
    head = nw._dense(256, 40).to(DEV)
    x = torch.randn((1200, 64), device=DEV, requires_grad=True)
    with nw.ln_bwd_links():
        y = head(blocks[1](blocks[0](x)))
    cot = torch.randn_like(y)
    params = [x] + [p for b in blocks for p in b.parameters()] + list(head.parameters())
    first = torch.autograd.grad(y, params, cot, retain_graph=True)
    second = torch.autograd.grad(y, params, cot)
    for a, b in zip(first, second):
        assert (a - b).abs().max() <= 2e-5 * (a.abs().max() + 1e-12), float((a - b).abs().max() / a.abs().max())


@pytest.mark.parametrize("M,widths", [(20480, (696, 256, 256, 1)), (2048, (696, 512, 512, 256, 1)), (333, (40, 24, 8, 1)), (5120, (696, 256, 256, 1))])
def test_value_net_silu_layers_without_torch_elementwise_kernels(M, widths):
    """brax value MLP (Dense -> SiLU ... Dense(1), track_mjx/agent/mlp_ppo/ppo_networks.py:180-184) on the GPU: forward one launch per hidden layer
    (tmjx_gemm_nt_silu), backward tmjx_silu_bwd + the MFMA input / weight gradient kernels; value and all gradients against float64 torch."""
    from track_mjx_amd.agent.networks import ValueNet
    torch.manual_seed(M)
    net = ValueNet(widths[0], widths[1:-1]).to(DEV)
    with torch.no_grad():
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(0.3 * torch.randn_like(m.bias))
    x = torch.randn((M, widths[0]), device=DEV)
    v = net(x)
    cot = torch.randn_like(v)
    params = list(net.parameters())
    grads = torch.autograd.grad(v, params, cot)
    ref = torch.nn.Sequential(*[torch.nn.Linear(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else torch.nn.SiLU() for m in net.net]).double().to(DEV)
    ref.load_state_dict({k: t.double() for k, t in net.net.state_dict().items()})
    vr = ref(x.double()).squeeze(-1)
    gr = torch.autograd.grad(vr, list(ref.parameters()), cot.double())
    assert (v.double() - vr).abs().max() <= 2e-5 * vr.abs().max() + 1e-6
    for a, b in zip(grads, gr):
        assert (a.double() - b).abs().max() <= 2e-4 * b.abs().max() + 1e-6, float((a.double() - b).abs().max() / b.abs().max())
    with torch.no_grad():
        assert torch.allclose(net(x), v, rtol=0, atol=0)          # the inference path runs the same kernels


@pytest.mark.parametrize("M,N,K", [(20480, 256, 256), (5120, 256, 256), (2048, 512, 512), (2048, 256, 512), (333, 24, 8), (1000, 128, 64), (77, 64, 40), (81, 100, 36)])
def test_input_gradient_with_silu_backward_epilogue_keeps_the_bits(M, N, K):
    """tmjx_gemm_nn_silu_bwd (k_gemm_act<.., EPI = 4>): dZ = (dY W) silu'(z + bias) in one launch against tmjx_gemm_nn followed by tmjx_silu_bwd — the same
    accumulators, the same expression: bit for bit; both tile heights (80 and 32 rows by M), ragged rows and widths."""
    from track_mjx_amd.agent import networks as nw
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dy = torch.randn((M, K), generator=g, device=DEV)
    w = torch.randn((K, N), generator=g, device=DEV) / K ** 0.5          # the consumer's weight [out = K][in = N]
    z = torch.randn((M, N), generator=g, device=DEV)
    b = 0.3 * torch.randn(N, generator=g, device=DEV)
    dx = nw.gemm_nn(dy, w)
    want = torch.empty_like(z)
    nw._launch("tmjx_silu_bwd", DEV, nw._p(dx), nw._p(z), nw._p(b), nw._p(want), M, N)
    got = torch.full_like(z, 7.0)
    nw._launch("tmjx_gemm_nn_silu_bwd", DEV, nw._p(dy), dy.stride(0), nw._p(w), w.stride(0), nw._p(z), nw._p(b), nw._p(got), M, N, K)
    torch.cuda.synchronize()
    assert torch.equal(got, want), float((got - want).abs().max())
    s = torch.sigmoid(z.double() + b.double())
    ref = (dy.double() @ w.double()) * (s * (1 + (z.double() + b.double()) * (1 - s)))
    assert (got.double() - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-6


@pytest.mark.parametrize("M,N", [(20480, 256), (5117, 512), (333, 8), (1, 4)])
def test_rank1_silu_backward_and_head_gradients(M, N):
    """tmjx_silu_bwd_rank1 against the outer product through tmjx_gemm_nn + tmjx_silu_bwd (bit for bit: one product, the same expression) and tmjx_head_dw
    against float64 (a sum over rows in another order than tmjx_gemm_dw's slabs: the fp32 dot-product bound)."""
    from track_mjx_amd import hip
    from track_mjx_amd.agent import networks as nw
    g = torch.Generator(device=DEV).manual_seed(M + N)
    dy1 = torch.randn(M, generator=g, device=DEV)
    w1 = torch.randn((1, N), generator=g, device=DEV)
    z = torch.randn((M, N), generator=g, device=DEV)
    b = 0.3 * torch.randn(N, generator=g, device=DEV)
    dx = nw.gemm_nn(dy1.view(M, 1), w1)
    want = torch.empty_like(z)
    nw._launch("tmjx_silu_bwd", DEV, nw._p(dx), nw._p(z), nw._p(b), nw._p(want), M, N)
    got = torch.full_like(z, 7.0)
    nw._launch("tmjx_silu_bwd_rank1", DEV, nw._p(dy1), nw._p(w1), nw._p(z), nw._p(b), nw._p(got), M, N)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # head gradients: x with a leading dimension
    buf = torch.randn((M, N + 4), generator=g, device=DEV)
    x = buf[:, :N]
    dw, db = torch.full((1, N), 7.0, device=DEV), torch.full((1,), 7.0, device=DEV)
    scratch = torch.empty(int(hip.lib().tmjx_head_dw_scratch_floats(M, N)), device=DEV)
    nw._launch("tmjx_head_dw", DEV, nw._p(dy1), nw._p(x), x.stride(0), nw._p(dw), nw._p(db), nw._p(scratch), M, N)
    torch.cuda.synchronize()
    ref = dy1.double() @ x.double()
    bound = (dy1.double().abs() @ x.double().abs()) * EPS * (M ** 0.5 + 4) * 2 + 1e-30
    assert ((dw[0].double() - ref).abs() <= bound).all()
    assert abs(float(db[0]) - float(dy1.double().sum())) <= float(dy1.double().abs().sum()) * EPS * (M ** 0.5 + 4) * 2 + 1e-30


@pytest.mark.parametrize("M,widths", [(20480, (696, 256, 256, 1)), (2048, (696, 512, 512, 256, 1)), (333, (40, 24, 8, 1))])
def test_value_chain_function_against_the_layer_by_layer_path(M, widths, monkeypatch):
    """ValueNet's learner pass as ONE autograd function (_ValueChainFn) against the per-layer functions (TMJX_VALUE_CHAIN=0): value and the hidden layers'
    gradients bit for bit (same kernels forward, same expressions backward), the 1-wide head's gradients to the dot-product bound; under
    deferred_weight_grads with flat gradient views the hidden layers' weight gradients land in the views."""
    from track_mjx_amd.agent import networks as nw
    torch.manual_seed(M + 1)
    net = nw.ValueNet(widths[0], widths[1:-1]).to(DEV)
    with torch.no_grad():
        for m in net.net:
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(0.3 * torch.randn_like(m.bias))
    x = torch.randn((M, widths[0]), device=DEV)
    cot = torch.randn(M, device=DEV)
    params = list(net.parameters())
    seen = []
    orig, orig_f32 = nw._ValueChainFn.forward, nw._F32ChainFn.forward
    monkeypatch.setattr(nw._ValueChainFn, "forward", staticmethod(lambda *a: (seen.append(1), orig(*a))[1]))
    # (a critic whose hidden layers are all 256 wide takes the whole-chain kernels, csrc/mlp_chain.h — one launch each way, bit-identical too: tests/test_gpu_chain.py)
    monkeypatch.setattr(nw._F32ChainFn, "forward", staticmethod(lambda *a: (seen.append(1), orig_f32(*a))[1]))
    v1 = net(x)
    g1 = torch.autograd.grad(v1, params, cot)
    assert seen, "the fused value chain did not run"
    monkeypatch.setenv("TMJX_VALUE_CHAIN", "0")
    n0 = len(seen)
    v0 = net(x)
    g0 = torch.autograd.grad(v0, params, cot)
    assert len(seen) == n0
    assert torch.equal(v0, v1)
    for i, (a, b) in enumerate(zip(g1, g0)):
        if i < len(params) - 2:
            assert torch.equal(a, b), (i, float((a - b).abs().max()))
        else:
            assert (a - b).abs().max() <= 2e-5 * b.abs().max() + 1e-6, (i, float((a - b).abs().max()))
    monkeypatch.delenv("TMJX_VALUE_CHAIN")
    # with flat gradient views: the deferred group writes the hidden layers' gradients there
    for p in params:
        p.grad = torch.full_like(p, 7.0)
    with nw.deferred_weight_grads() as d:
        g2 = torch.autograd.grad(net(x), params, cot)
        d.launch()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(g2, g0)):
        assert (a - b).abs().max() <= 2e-5 * b.abs().max() + 1e-6, (i, float((a - b).abs().max()))
    assert g2[0].data_ptr() == params[0].grad.data_ptr()
