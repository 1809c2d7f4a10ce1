"""GPU: the whole-chain kernels (csrc/mlp_chain.h through the C-ABI: tmjx_chain_fwd / tmjx_chain_bwd) against

  * the layer-by-layer entry points they replace (tmjx_gemm_nt_silu_ln, tmjx_gemm_nt_silu, tmjx_gemm_nt, tmjx_head_fwd, tmjx_gemm_nn_ln_bwd,
    tmjx_gemm_nn): BIT-IDENTICAL — every output element is the same k-ordered fp32 MFMA chain and the epilogues are the same expressions;
  * plain fp32 / float64 torch on the same weights (Dense -> silu -> LayerNorm blocks of track_mjx/agent/mlp_ppo/intention_network.py:32-44,68-76,
    brax's value MLP ppo_networks.py:180-184), tolerance 2e-5 relative to the row's magnitude (fp32 accumulation over K <= 696).

Row counts: 20 480 (BASELINE configs[1]: 1024 minibatch rows x unroll 20, the 80-row tile), 5 120 (configs[2]'s rank share, the 32-row tile), 1 365 (one
env group of the roll-out), ragged sizes that end inside a tile."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(g, K0, n_hidden, Nf, kind):
    hidden, k = [], K0
    ld = (K0 + 3) // 4 * 4                # the learner's flat parameter buffers pad weight rows to multiples of 4 floats (470 -> 472, 286 -> 288)
    for _ in range(n_hidden):
        wbuf = torch.zeros((256, ld if k == K0 else 256), device=DEV)
        wbuf[:, :k] = torch.randn((256, k), generator=g, device=DEV) / k ** 0.5
        w = wbuf[:, :k]
        b = torch.randn(256, generator=g, device=DEV) * 0.1
        if kind == "ln":
            hidden.append((w, b, 1 + 0.1 * torch.randn(256, generator=g, device=DEV), 0.1 * torch.randn(256, generator=g, device=DEV)))
        else:
            hidden.append((w, b))
        k = 256
    final = None
    if Nf:
        final = (torch.randn((Nf, 256), generator=g, device=DEV) / 16.0, torch.randn(Nf, generator=g, device=DEV) * 0.1)
    return hidden, final


def _layer_by_layer(x2, hidden, final, kind, eps=1e-6):
    from track_mjx_amd.agent.networks import _launch, _p, gemm_nt
    M = x2.shape[0]
    saved, h = [], x2
    for lay in hidden:
        w, b = lay[0], lay[1]
        z = torch.empty((M, 256), device=DEV); y = torch.empty_like(z)
        if kind == "ln":
            st = torch.empty((M, 2), device=DEV)
            _launch("tmjx_gemm_nt_silu_ln", DEV, _p(h), h.stride(0), _p(w), w.stride(0), _p(b), _p(lay[2]), _p(lay[3]), _p(z), _p(y), 256, _p(st), M, 256, w.shape[1], float(eps))
        else:
            st = None
            _launch("tmjx_gemm_nt_silu", DEV, _p(h), h.stride(0), _p(w), w.stride(0), _p(b), _p(z), _p(y), 256, M, 256, w.shape[1])
        saved.append((z, y, st))
        h = y
    out = None
    if final is not None:
        wf, bf = final
        if wf.shape[0] == 1:
            out = torch.empty(M, device=DEV)
            _launch("tmjx_head_fwd", DEV, _p(h), h.stride(0), _p(wf), _p(bf), _p(out), M, 256)
        else:
            out = gemm_nt(h, wf, bf)
    return saved, out


def _torch_ref(x2, hidden, final, kind, eps=1e-6):
    h = x2.double()
    for lay in hidden:
        v = torch.nn.functional.silu(h @ lay[0].double().t() + lay[1].double())
        h = torch.nn.functional.layer_norm(v, (256,), lay[2].double(), lay[3].double(), eps) if kind == "ln" else v
    if final is not None:
        h = h @ final[0].double().t() + final[1].double()
        if final[0].shape[0] == 1:
            h = h[:, 0]
    return h


CASES = [  # (rows, K0, lda, hidden layers, Nf, kind)
    (20480, 470, 696, 2, 120, "ln"),      # encoder + fc2_mean | fc2_logvar at the headline configuration
    (20480, 286, 288, 2, 76, "ln"),       # decoder + action head
    (20480, 696, 696, 2, 1, "silu"),      # critic
    (5120, 470, 696, 2, 120, "ln"), (5120, 286, 288, 2, 76, "ln"), (5120, 696, 696, 2, 1, "silu"),      # one rank's share of BASELINE configs[2]: 32-row tiles
    (1365, 470, 696, 2, 120, "ln"), (1368, 696, 696, 2, 1, "silu"),                                      # one env group of the roll-out
    (1024, 696, 696, 2, 1, "silu"),       # the bootstrap value of a minibatch
    (1000, 470, 696, 1, 120, "ln"), (77, 286, 288, 3, 76, "ln"), (20481, 256, 256, 4, 0, "ln"), (333, 64, 64, 2, 0, "silu"), (1, 696, 696, 2, 1, "silu"),
    (95, 470, 472, 2, 128, "ln"), (4099, 100, 100, 2, 5, "silu"),
]


@pytest.mark.parametrize("M,K0,lda,nh,Nf,kind", CASES)
def test_forward_chain_is_bit_identical_to_the_layer_by_layer_kernels(M, K0, lda, nh, Nf, kind):
    from track_mjx_amd.agent.networks import chain_fwd, chain_fwd_ok
    g = torch.Generator(device=DEV).manual_seed(M + K0 + nh + Nf)
    x2 = torch.randn((M, lda), generator=g, device=DEV)[:, :K0]
    hidden, final = _net(g, K0, nh, Nf, kind)
    assert chain_fwd_ok(x2, hidden, final, kind)
    saved, out = chain_fwd(x2, hidden, final, kind)
    saved_r, out_r = _layer_by_layer(x2, hidden, final, kind)
    torch.cuda.synchronize()
    for l, ((z, y, st), (zr, yr, sr)) in enumerate(zip(saved, saved_r)):
        assert torch.equal(z, zr), (l, float((z - zr).abs().max()))
        assert torch.equal(y, yr), (l, float((y - yr).abs().max()))
        if kind == "ln":
            assert torch.equal(st, sr), l
    if final is not None:
        assert out.shape == out_r.shape and torch.equal(out, out_r), float((out - out_r).abs().max())
    ref = _torch_ref(x2, hidden, final, kind)
    got = (out if final is not None else saved[-1][1]).double()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())


def test_forward_chain_rejects_what_it_cannot_run():
    from track_mjx_amd import hip
    from track_mjx_amd.agent.networks import chain_fwd_desc, chain_fwd_ok
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn((64, 472), generator=g, device=DEV)
    hidden, final = _net(g, 470, 2, 120, "ln")
    assert chain_fwd_ok(x[:, :470], hidden, final, "ln")
    assert not chain_fwd_ok(x[:, 1:471], hidden, final, "ln")                  # rows not 16-byte aligned
    wide = [(torch.randn((512, 470), device=DEV),) + hidden[0][1:]] + hidden[1:]
    assert not chain_fwd_ok(x[:, :470], wide, final, "ln")                     # a hidden layer that is not 256 wide
    big = (torch.randn((130, 256), device=DEV), torch.randn(130, device=DEV))
    assert not chain_fwd_ok(x[:, :470], hidden, big, "ln")                     # a last layer wider than 128
    d, _, _ = chain_fwd_desc(x[:, :470], hidden, big, "ln")
    assert hip.lib().tmjx_chain_fwd(C.byref(d), None) == -22 and b"128" in hip.lib().tmjx_last_error()


def _bwd_layer_by_layer(g, final_w, blocks, kind, w0, dx_cols):
    """The launches the backward chain replaces: per hidden layer tmjx_gemm_nn_ln_bwd (LayerNorm blocks) / tmjx_gemm_nn_silu_bwd, the value head's
    tmjx_silu_bwd_rank1 in front, tmjx_gemm_nn for the trailing input gradient."""
    from track_mjx_amd import hip
    from track_mjx_amd.agent.networks import _launch, _p, gemm_nn
    L = hip.lib()
    M = g.shape[0]
    dzs, partials = [], []
    cur = g
    for i, blk in enumerate(blocks):
        w = final_w if i == 0 else blocks[i - 1][0]
        dz = torch.empty((M, 256), device=DEV)
        if kind == "ln":
            pt = torch.empty(int(L.tmjx_gemm_nn_ln_bwd_partial_floats(M, 256)), device=DEV)
            _launch("tmjx_gemm_nn_ln_bwd", DEV, _p(cur), cur.stride(0), _p(w), w.stride(0), _p(blk[1]), _p(blk[2]), _p(blk[3]), _p(blk[4]), _p(dz), _p(pt), M, 256, cur.shape[1])
            partials.append(pt)
        elif i == 0 and g.dim() == 1:
            _launch("tmjx_silu_bwd_rank1", DEV, _p(g), _p(final_w), _p(blk[1]), _p(blk[2]), _p(dz), M, 256)
        else:
            _launch("tmjx_gemm_nn_silu_bwd", DEV, _p(cur), cur.stride(0), _p(w), w.stride(0), _p(blk[1]), _p(blk[2]), _p(dz), M, 256, cur.shape[1])
        dzs.append(dz)
        cur = dz
    dx = gemm_nn(cur, w0, dx_cols) if w0 is not None else None
    return dzs, partials, dx


BWD_CASES = [  # (rows, Kg, hidden layers, kind, K0, dx_cols)
    (20480, 76, 2, "ln", 286, 60),        # decoder: action head -> two blocks -> d loss / d latent
    (20480, 120, 2, "ln", 470, None),     # encoder: fc2 -> two blocks
    (20480, 1, 2, "silu", 696, None),     # critic: 1-wide head -> two layers
    (5120, 76, 2, "ln", 286, 60), (5120, 120, 2, "ln", 470, None), (5120, 1, 2, "silu", 696, None),
    (1000, 76, 1, "ln", 286, 60), (77, 120, 3, "ln", 470, None), (333, 1, 1, "silu", 64, None), (4099, 64, 2, "silu", 100, 33), (95, 128, 4, "ln", 256, 128),
    (1, 1, 3, "silu", 696, None),
]


@pytest.mark.parametrize("M,Kg,nh,kind,K0,dx_cols", BWD_CASES)
def test_backward_chain_is_bit_identical_to_the_layer_by_layer_kernels(M, Kg, nh, kind, K0, dx_cols):
    from track_mjx_amd.agent.networks import chain_bwd, chain_fwd
    g_ = torch.Generator(device=DEV).manual_seed(7 * M + Kg + nh)
    lda = (K0 + 3) // 4 * 4
    x2 = torch.randn((M, lda), generator=g_, device=DEV)[:, :K0]
    hidden, final = _net(g_, K0, nh, Kg, kind)
    saved, out = chain_fwd(x2, hidden, final, kind)                    # a real forward pass provides z / stats
    g = torch.randn((M,) if Kg == 1 else (M, Kg), generator=g_, device=DEV)
    if kind == "ln":
        blocks = [(hidden[l][0], saved[l][0], hidden[l][1], hidden[l][2], saved[l][2]) for l in range(nh - 1, -1, -1)]
    else:
        blocks = [(hidden[l][0], saved[l][0], hidden[l][1]) for l in range(nh - 1, -1, -1)]
    w0 = hidden[0][0] if dx_cols else None
    dzs, partials, dx = chain_bwd(g, final[0], blocks, kind, w0, dx_cols)
    dzs_r, partials_r, dx_r = _bwd_layer_by_layer(g, final[0], blocks, kind, w0, dx_cols)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(dzs, dzs_r)):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    if kind == "ln":
        for i, (a, b) in enumerate(zip(partials, partials_r)):
            assert torch.equal(a, b), i
    if dx_cols:
        assert torch.equal(dx[:, :dx_cols], dx_r[:, :dx_cols]), float((dx[:, :dx_cols] - dx_r[:, :dx_cols]).abs().max())
    # against float64 autograd on the same weights: d loss / d z of the FIRST hidden layer (everything upstream enters it)
    xs = x2.double().requires_grad_(True)
    h, zs = xs, []
    for lay in hidden:
        z = h @ lay[0].double().t()
        z.retain_grad(); zs.append(z)
        v = torch.nn.functional.silu(z + lay[1].double())
        h = torch.nn.functional.layer_norm(v, (256,), lay[2].double(), lay[3].double(), 1e-6) if kind == "ln" else v
    o = h @ final[0].double().t() + final[1].double()
    (o * (g.double()[:, None] if Kg == 1 else g.double())).sum().backward()
    ref = zs[0].grad
    got = dzs[-1].double()
    assert (got - ref).abs().max() <= 5e-5 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    if dx_cols:
        assert (dx[:, :dx_cols].double() - xs.grad[:, :dx_cols]).abs().max() <= 5e-5 * max(1.0, float(xs.grad.abs().max()))


def _learner_2x256(rows_per_minibatch, seed_env=0):
    """The headline nets (encoder / decoder / critic = [256, 256]: BASELINE configs[1] / configs[2]) on a small env batch: 256 envs, unroll 20,
    `rows_per_minibatch` rows x 20 steps per SGD step."""
    from tests.common import make_env_and_oracle
    from track_mjx_amd.agent import ppo
    envs = [make_env_and_oracle(num_envs=128, n_clips=4, wrappers=True, seed=seed_env + k)[0] for k in range(2)]
    L = ppo.PPOLearner(envs, encoder_layers=(256, 256), decoder_layers=(256, 256), critic_layers=(256, 256), latents=60, unroll_length=20,
                       batch_size=rows_per_minibatch, num_minibatches=256 // rows_per_minibatch, num_updates_per_batch=2, seed=3, normalize_observations=True)
    for k, e in enumerate(envs):
        L.states[k] = e.reset(torch.Generator().manual_seed(10 + k))
    return L


@pytest.mark.parametrize("rows,latent_tail", [(128, False), (64, False), (128, True)])
def test_training_with_the_chain_kernels_is_bit_identical_to_the_layer_by_layer_learner(rows, latent_tail, monkeypatch):
    """Two training steps (roll-out, normaliser, 2 x (256 / rows) minibatch SGD steps each, captured as hipGraphs) of the 2 x 256 nets with the whole-chain
    kernels and with TMJX_NO_CHAIN=1: same parameters, Adam moments and metrics to the bit — and the chain path really ran (k_chain_* entry points counted)."""
    from track_mjx_amd.agent import networks
    calls = {"fwd": 0, "bwd": 0}
    real_fwd, real_bwd = networks.chain_fwd, networks.chain_bwd

    def cf(*a, **k):
        calls["fwd"] += 1
        return real_fwd(*a, **k)

    def cb(*a, **k):
        calls["bwd"] += 1
        return real_bwd(*a, **k)
    monkeypatch.setattr(networks, "chain_fwd", cf)
    monkeypatch.setattr(networks, "chain_bwd", cb)
    if latent_tail:        # the encoder launch also samples the latent and writes the decoder's input (off by default: measured slower, networks.py)
        monkeypatch.setenv("TMJX_CHAIN_LATENT", "1")
    L = _learner_2x256(rows)
    ms = [L.training_step(it) for it in range(2)]
    torch.cuda.synchronize()
    assert calls["fwd"] >= 4 and calls["bwd"] >= 3, calls      # (captured once into the SGD-step graph: encoder, decoder, critic, bootstrap value | three backward chains)
    got = (L.opt.flat.clone(), L.opt.exp_avg.clone(), L.opt.exp_avg_sq.clone())
    monkeypatch.setenv("TMJX_NO_CHAIN", "1")
    before = dict(calls)
    R = _learner_2x256(rows)
    mr = [R.training_step(it) for it in range(2)]
    torch.cuda.synchronize()
    assert calls == before, "TMJX_NO_CHAIN=1 must take the layer-by-layer functions"
    assert torch.equal(got[0], R.opt.flat) and torch.equal(got[1], R.opt.exp_avg) and torch.equal(got[2], R.opt.exp_avg_sq)
    for a, b in zip(ms, mr):
        assert a.keys() == b.keys() and all(torch.equal(torch.as_tensor(a[k]), torch.as_tensor(b[k])) for k in a), (a, b)


@pytest.mark.parametrize("M", [20480, 5120, 1365, 77])
def test_latent_tail_of_the_encoder_chain_equals_tmjx_latent_concat(M):
    """tmjx_chain_fwd with lat_out: the encoder launch also samples the latent and writes the decoder's input [mean + eps exp(logvar / 2) | obs[:, 470:]]
    (reparameterize + concat, intention_network.py:84-88,128-139) — bit for bit what the separate tmjx_latent_concat launch writes from the same fc2."""
    from track_mjx_amd.agent.networks import _launch, _p, chain_fwd
    g = torch.Generator(device=DEV).manual_seed(M)
    obs = torch.randn((M, 696), generator=g, device=DEV)
    hidden, final = _net(g, 470, 2, 120, "ln")
    eps = torch.randn((M, 60), generator=g, device=DEV)
    dec_in = torch.full((M, 288), 7.0, device=DEV)
    saved, fc2 = chain_fwd(obs[:, :470], hidden, final, "ln", latent=(eps, dec_in, obs[:, 470:]))
    saved_r, fc2_r = chain_fwd(obs[:, :470], hidden, final, "ln")
    ref = torch.full((M, 288), 7.0, device=DEV)
    _launch("tmjx_latent_concat", DEV, _p(fc2_r), _p(eps), _p(obs), _p(ref), M, 60, 696, 470, 696, 1, None, None, 288, 0, None)
    torch.cuda.synchronize()
    assert torch.equal(fc2, fc2_r) and all(torch.equal(a[1], b[1]) for a, b in zip(saved, saved_r))
    assert torch.equal(dec_in[:, :286], ref[:, :286]), float((dec_in[:, :286] - ref[:, :286]).abs().max())
    want = fc2_r[:, :60].double() + eps.double() * torch.exp(0.5 * fc2_r[:, 60:].double())
    assert (dec_in[:, :60].double() - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max()))
