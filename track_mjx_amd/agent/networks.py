"""Actor / critic networks of the PPO learner.

Mirrors (paths relative to /root/reference/track_mjx/agent/mlp_ppo):
  intention_network.py:14-142   Encoder / Decoder / reparameterize / IntentionNetwork
                                (Dense -> SiLU -> LayerNorm per hidden layer: the LayerNorm comes AFTER the
                                activation; fc2_mean / fc2_logvar heads; decoder's last Dense is un-activated)
  ppo_networks.py:34-100,157-190  make_inference_fn policy (sample / log_prob / postprocess) and
                                make_intention_ppo_networks (value net = brax make_value_network: swish MLP -> 1)
  brax.training.distribution.NormalTanhDistribution (third-party, restated from its published definition):
                                loc, raw = split(logits); scale = softplus(raw) + 0.001; tanh bijector
The dense contractions run on the matrix cores through the library's own kernels (csrc/gemm_kernels.h: tmjx_gemm_nt / _nn / _dw,
fp32 in / fp32 accumulate v_mfma_f32_16x16x4_f32, 80-row tiles = one workgroup per CU at 20 480 rows, bias gradient fused into the
weight-gradient kernel) — forward, input gradient and weight gradient alike, under autograd and in inference.  Everything is fp32 as
in the reference unless `matmul_dtype=torch.bfloat16` is requested (BASELINE config 5): then the GEMM INPUTS are bf16
(`gemm_inputs` + `Bf16Shadows`: the library's own v_mfma_f32_16x16x32_bf16 kernels, csrc/gemm_bf16.h), accumulation, outputs and
everything around the GEMMs stay fp32.  CPU tensors (host-logic tests) take plain torch ops.
"""
from __future__ import annotations

import math
import os
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


def _lecun_uniform_(w: torch.Tensor) -> None:
    fan_in = w.shape[1]
    lim = math.sqrt(3.0 / fan_in)
    with torch.no_grad():
        w.uniform_(-lim, lim)


def _lecun_normal_(w: torch.Tensor) -> None:
    fan_in = w.shape[1]
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
    with torch.no_grad():
        nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std)


class gemm_inputs:
    """`with gemm_inputs(torch.bfloat16, shadows):` — the dense contractions of the networks take bf16 INPUTS with fp32 accumulation and
    fp32 OUTPUTS on the library's own v_mfma_f32_16x16x32_bf16 kernels (csrc/gemm_bf16.h: activations converted when they are staged into
    LDS, weights from the bf16 `shadows` of the fp32 master parameters — no cast pass, no library GEMM); parameters, the fused
    epilogues, the loss head and the optimiser stay fp32 (BASELINE config 5).  No autocast.  None = fp32.  CPU tensors (host-logic tests)
    take torch ops in fp32."""
    dtype = None
    shadows = None          # Bf16Shadows of the networks' dense layers (CUDA: the library's own bf16 kernels take every contraction)
    # bf16 TWINS of fp32 network inputs, made by the kernel that wrote the input (tmjx_minibatch_begin_bf16: the normalised observation rounded to
    # nearest even, [rows][ld >= ceil64(width)], zero beyond the width): {data_ptr of the fp32 tensor: (twin [rows, ld], width)}.  A chain whose
    # input starts at a registered pointer stages the twin instead — the bits its GEMMs would have made of the fp32 rows themselves, half the
    # bytes, and the LDS-DMA form of the kernels (whole 64-column K tiles: the twin's columns beyond the layer's K meet zero weight columns)
    twins = None

    def __init__(self, dtype, shadows=None, twins=None):
        self.new = (dtype, shadows if dtype is not None else None, twins if dtype is not None else None)

    def __enter__(self):
        self.old = (gemm_inputs.dtype, gemm_inputs.shadows, gemm_inputs.twins)
        gemm_inputs.dtype, gemm_inputs.shadows, gemm_inputs.twins = self.new

    def __exit__(self, *a):
        gemm_inputs.dtype, gemm_inputs.shadows, gemm_inputs.twins = self.old
        return False

    @staticmethod
    def twin_of(x2: torch.Tensor, K: int):
        """The bf16 twin's first ceil64(K) columns for the fp32 rows `x2` (a column prefix of a registered tensor), or None."""
        tw = gemm_inputs.twins
        if not tw or x2.dtype != torch.float32:
            return None
        hit = tw.get(x2.data_ptr())
        if hit is None:
            return None
        t16, width = hit
        Ke = _ceil64(K)
        if t16.shape[0] != x2.shape[0] or x2.stride(0) != width or K > width or Ke > t16.shape[1] or os.environ.get("TMJX_NO_BF16_TWIN"):
            return None
        return t16[:, :Ke]


def _hip_gemm_ok(*ts) -> bool:
    """fp32 CUDA operands and no reduced-precision GEMM-input mode: the hand-written fp32 MFMA kernels take the contraction."""
    return gemm_inputs.dtype is None and all(t.is_cuda and t.dtype == torch.float32 for t in ts)


def _rows2d(x: torch.Tensor) -> torch.Tensor:
    """[.., K] -> [rows, K] with unit column stride (a view whenever the leading dims collapse, e.g. the [T, B, :470] slice of the
    observation buffer keeps its row stride of 696: the kernels take a leading dimension, no copy)."""
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.stride(1) == 1 and (x2.shape[0] == 1 or x2.stride(0) >= x2.shape[1]) else x2.contiguous()


def _launch(fn_name: str, dev, *args):
    import ctypes as C
    from .. import hip as _hip
    L = _hip.lib()
    with torch.cuda.device(dev):
        _hip.check(getattr(L, fn_name)(*args, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), fn_name)


def _p(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr()) if t is not None else None


def gemm_nt(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y[rows, N] = x[rows, K] w[N, K]^T + bias  (tmjx_gemm_nt)."""
    x = _rows2d(x)
    w = w if w.stride(1) == 1 else w.contiguous()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if N == 1 and _head_kernels() and K % 4 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0:
        # a 1-wide layer (the value MLP's head) is a matrix-vector product: a wave per row instead of a 16 x 64 MFMA tile at 1 / 64 of its width — in EVERY
        # pass that computes it (learner, bootstrap value, evaluation), so that they agree to the bit
        _launch("tmjx_head_fwd", x.device, _p(x), x.stride(0), _p(w), _p(bias), _p(y), M, K)
        return y
    _launch("tmjx_gemm_nt", x.device, _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(y), N, M, N, K)
    return y


def _head_kernels() -> bool:
    return os.environ.get("TMJX_HEAD_KERNELS", "1") != "0"


def gemm_nn(dy: torch.Tensor, w: torch.Tensor, cols: int | None = None) -> torch.Tensor:
    """dx[rows, K] = dy[rows, N] w[N, K]  (tmjx_gemm_nn); `cols`: only the first `cols` columns of dx are computed (the rest of the
    [rows, K] buffer is left unwritten: a caller that needs the gradient of a column prefix only)."""
    dy = _rows2d(dy)
    w = w if w.stride(1) == 1 else w.contiguous()
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    _launch("tmjx_gemm_nn", dy.device, _p(dy), dy.stride(0), _p(w), w.stride(0), _p(dx), K, M, K if cols is None else int(cols), N)
    return dx


def gemm_dw(dy: torch.Tensor, x: torch.Tensor, with_bias: bool):
    """(dw[N, K], db[N] or None) = (dy^T x, column sums of dy)  (tmjx_gemm_dw: row-range slabs + one reduction launch)."""
    from .. import hip as _hip
    dy, x = _rows2d(dy), _rows2d(x)
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if with_bias else None
    if N == 1 and _head_kernels() and dy.stride(0) == 1:
        # a 1-wide layer's gradients: a matrix-vector product (tmjx_head_dw) instead of a 128 x 128 tile at 1 / 128 of its width
        scratch = torch.empty(int(_hip.lib().tmjx_head_dw_scratch_floats(M, K)), dtype=torch.float32, device=dy.device)
        _launch("tmjx_head_dw", dy.device, _p(dy), _p(x), x.stride(0), _p(dw), _p(db), _p(scratch), M, K)
        return dw, db
    scratch = torch.empty(int(_hip.lib().tmjx_gemm_dw_scratch_floats(M, N, K)), dtype=torch.float32, device=dy.device)
    _launch("tmjx_gemm_dw", dy.device, _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), _p(db), _p(scratch), M, N, K)
    return dw, db


# ---- bf16 GEMM-input mode on the library's own kernels (csrc/gemm_bf16.h) -------------------------------------------------------------
def _ceil64(n: int) -> int:
    return (int(n) + 63) // 64 * 64


def _ld(x2: torch.Tensor) -> int:
    """Leading dimension of a [rows, cols] operand for the bf16 kernels (a one-row tensor's stride is arbitrary: any aligned value >= cols)."""
    return x2.stride(0) if x2.shape[0] > 1 else (x2.shape[1] + 7) // 8 * 8


def bgemm_nt(x: torch.Tensor, wb: torch.Tensor, N: int, K: int, bias: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """y[rows, N] = x[rows, K] wb[N, K]^T + bias on v_mfma_f32_16x16x32_bf16 (tmjx_bgemm_nt): x fp32 (converted to bf16 when it is staged) or
    bf16; wb = a bf16 shadow [>= N rows][>= ceil64(K)], zero beyond K; fp32 result.  `out`: a [rows, >= N] buffer whose first N columns are written."""
    x = _rows2d(x)
    M = x.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device) if out is None else out
    _launch("tmjx_bgemm_nt", x.device, _p(x), int(x.dtype == torch.float32), _ld(x), _p(wb), wb.stride(0), _p(bias), _p(y), max(y.stride(0), N), M, N, K)
    return y


def bgemm_dw(dy: torch.Tensor, x: torch.Tensor, with_bias: bool, out: torch.Tensor | None = None, out_bias: torch.Tensor | None = None):
    """(dw[N, K], db[N] or None) = (dy^T x, column sums of dy) with bf16 operands (tmjx_bgemm_dw: transposed LDS reads, row-range slabs + one
    reduction launch); dy / x fp32 or bf16.  `out` / `out_bias`: destinations (e.g. flat-buffer gradient views; `out` may be row-padded)."""
    from .. import hip as _hip
    dy, x = _rows2d(dy), _rows2d(x)
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device) if out is None else out
    db = (torch.empty(N, dtype=torch.float32, device=dy.device) if out_bias is None else out_bias) if with_bias else None
    d = deferred_weight_grads.active
    if d is not None and d.group_bf16 and N > 1 and out is not None:
        # inside the learner's backward pass, into a flat-buffer view (the parameter's FIRST use in this pass: `dest` / deferred_weight_grads.seen — a second
        # use's gradient is added to the first by autograd DURING the pass and must exist by then): recorded, computed with every other layer's by ONE
        # grouped launch behind the pass (deferred_weight_grads.launch)
        d.bproblems.append((dy, x, dw, db))
        return dw, db
    scratch = torch.empty(int(_hip.lib().tmjx_bgemm_dw_scratch_floats(M, N, K)), dtype=torch.float32, device=dy.device)
    _launch("tmjx_bgemm_dw", dy.device, _p(dy), int(dy.dtype == torch.float32), _ld(dy), _p(x), int(x.dtype == torch.float32), _ld(x),
            _p(dw), dw.stride(0) if N > 1 else max(dw.stride(0), K), _p(db), _p(scratch), M, N, K)
    return dw, db


def bf16_z_dtype() -> torch.dtype:
    """Element type of the pre-activations the bf16-mode block kernels save for their backward pass: bf16 in the product build (BASELINE configs[4]:
    "bf16 MLP on MFMA"), fp32 in a library built with -DTMJX_BF16_Z_F32 (SURVEY a16's "bf16 only for GEMM inputs") — tmjx_bf16_z_bytes()."""
    from .. import hip as _hip
    return torch.float32 if int(_hip.lib().tmjx_bf16_z_bytes()) == 4 else torch.bfloat16


def bgemm_ln_fwd(x, wb, N: int, K: int, bias, gamma, beta, eps: float):
    """Dense -> SiLU -> LayerNorm forward in one launch (tmjx_bgemm_ln_fwd; N = 128 / 256 / 512): returns (z bf16 [rows, N] without the bias,
    y bf16 [rows, N], stats fp32 [rows, 2] = (mean, 1 / std))."""
    x = _rows2d(x)
    M = x.shape[0]
    z = torch.empty((M, N), dtype=bf16_z_dtype(), device=x.device)
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    _launch("tmjx_bgemm_ln_fwd", x.device, _p(x), int(x.dtype == torch.float32), _ld(x), _p(wb), wb.stride(0), _p(bias), _p(gamma), _p(beta), _p(z), N, _p(y), N,
            _p(stats), M, N, K, float(eps))
    return z, y, stats


def bgemm_ln_bwd(dy, wtb, N: int, K: int, z, bias, gamma, stats):
    """The input gradient of a block's CONSUMER (dy [rows, K] = gradient of the consumer's output, wtb = the consumer's transposed shadow) with the
    block's LayerNorm + SiLU backward in the epilogue (tmjx_bgemm_ln_bwd): returns (dz bf16 [rows, N] = d loss / d z of the block, partial =
    per-row-tile column sums [tiles, 3, N])."""
    from .. import hip as _hip
    dy = _rows2d(dy)
    M = dy.shape[0]
    dz = torch.empty((M, N), dtype=torch.bfloat16, device=dy.device)
    partial = torch.empty(int(_hip.lib().tmjx_bgemm_partial_floats(M, N, 3)), dtype=torch.float32, device=dy.device)
    _launch("tmjx_bgemm_ln_bwd", dy.device, _p(dy), int(dy.dtype == torch.float32), _ld(dy), _p(wtb), wtb.stride(0), _p(z), z.stride(0), _p(bias), _p(gamma), _p(stats),
            _p(dz), N, _p(partial), M, N, K)
    return dz, partial.view(-1, 3, N)


def bgemm_silu_fwd(x, wb, N: int, K: int, bias, y_f32: bool = False):
    """Dense -> SiLU forward in one launch (tmjx_bgemm_silu_fwd): (z bf16 without the bias, y = silu(z + bias) as bf16 — or fp32 with y_f32)."""
    x = _rows2d(x)
    M = x.shape[0]
    ld = (N + 7) // 8 * 8
    z = torch.empty((M, ld), dtype=bf16_z_dtype(), device=x.device)[:, :N]
    y = torch.empty((M, ld), dtype=torch.float32 if y_f32 else torch.bfloat16, device=x.device)
    _launch("tmjx_bgemm_silu_fwd", x.device, _p(x), int(x.dtype == torch.float32), _ld(x), _p(wb), wb.stride(0), _p(bias), _p(z), ld,
            None if y_f32 else _p(y), ld, _p(y) if y_f32 else None, ld, M, N, K)
    return z, y[:, :N]


def bgemm_silu_bwd(dy, wtb, N: int, K: int, z, bias):
    """The consumer's input-gradient GEMM with the Dense -> SiLU block's backward in the epilogue (tmjx_bgemm_silu_bwd): (dz bf16 [rows, N], partial
    [tiles, N] = column sums of dz per row tile)."""
    from .. import hip as _hip
    dy = _rows2d(dy)
    M = dy.shape[0]
    ld = (N + 7) // 8 * 8
    dz = torch.empty((M, ld), dtype=torch.bfloat16, device=dy.device)
    partial = torch.empty(int(_hip.lib().tmjx_bgemm_partial_floats(M, N, 1)), dtype=torch.float32, device=dy.device)
    _launch("tmjx_bgemm_silu_bwd", dy.device, _p(dy), int(dy.dtype == torch.float32), _ld(dy), _p(wtb), wtb.stride(0), _p(z), z.stride(0), _p(bias), _p(dz), ld,
            _p(partial), M, N, K)
    return dz[:, :N], partial.view(-1, N)


class Bf16Shadows:
    """bf16 copies of the dense layers' weights: `w[lin]` = [N][ceil64 K] (forward operand), `wt[lin]` = [K][ceil64 N] (input-gradient operand),
    zero padded, all refreshed by ONE launch (tmjx_bf16_shadow) — at the start of every SGD step, i.e. once per optimiser step."""

    def __init__(self, linears, need_t=None):
        from .. import hip as _hip
        self.lins = list(linears)
        need_t = set(self.lins if need_t is None else need_t)
        dev = self.lins[0].weight.device
        self.w, self.wt = {}, {}
        items = []
        for lin in self.lins:
            N, K = lin.weight.shape
            self.w[lin] = torch.zeros((N, _ceil64(K)), dtype=torch.bfloat16, device=dev)
            if lin in need_t:
                self.wt[lin] = torch.zeros((K, _ceil64(N)), dtype=torch.bfloat16, device=dev)
            t = self.wt.get(lin)
            items.append(_hip.Bf16Shadow(lin.weight.data_ptr(), self.w[lin].data_ptr(), t.data_ptr() if t is not None else None, N, K, lin.weight.stride(0),
                                         self.w[lin].stride(0), t.stride(0) if t is not None else 0))
        self._chunks = [(_hip.Bf16Shadow * len(items[i:i + 24]))(*items[i:i + 24]) for i in range(0, len(items), 24)]
        self._ptrs = [lin.weight.data_ptr() for lin in self.lins]

    def refresh(self) -> None:
        if self._ptrs != [lin.weight.data_ptr() for lin in self.lins]:
            raise RuntimeError("a parameter moved after its bf16 shadow was built (build the shadows after the flat optimiser re-seats the parameters)")
        dev = self.lins[0].weight.device
        for arr in self._chunks:
            _launch("tmjx_bf16_shadow", dev, arr, len(arr))


def _bf16_ok(x, lin) -> bool:
    """The bf16 kernels take this layer: bf16 GEMM-input mode with shadows, CUDA fp32 activations with 16-byte aligned rows, an output wide
    enough to have aligned gradient rows (the 1-wide value head stays on the fp32 kernels)."""
    sh = gemm_inputs.shadows
    return (gemm_inputs.dtype == torch.bfloat16 and sh is not None and lin in sh.w and x.is_cuda and x.dtype == torch.float32
            and lin.out_features % 4 == 0)


def _colsum(dy: torch.Tensor) -> torch.Tensor:
    """dy.sum(0) of a contiguous fp32 [rows, width] CUDA tensor through tmjx_colsum (two launches of ~4 us; torch's generic reduction
    needs 10-25 us for the tall, narrow bias-gradient inputs); anything else goes to torch."""
    if not (dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 2 and dy.is_contiguous() and dy.shape[0] >= 1024):
        return dy.sum(0)
    import ctypes as C
    from .. import hip as _hip
    L = _hip.lib()
    rows, width = dy.shape
    out = torch.empty(width, dtype=torch.float32, device=dy.device)
    scratch = torch.empty(L.tmjx_colsum_scratch_floats(width), dtype=torch.float32, device=dy.device)
    with torch.cuda.device(dy.device):
        _hip.check(L.tmjx_colsum(C.c_void_p(dy.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), rows, width,
                                 C.c_void_p(torch.cuda.current_stream(dy.device).cuda_stream)), "tmjx_colsum")
    return out


class deferred_weight_grads:
    """`with deferred_weight_grads() as d: grads = torch.autograd.grad(...)` then `d.launch()`: inside the block every dense layer's
    backward only RECORDS its weight-gradient problem (dY, X and the parameter's flat-buffer gradient views as destinations) and hands those
    views back to autograd as the gradients; `launch()` then computes all of them with ONE grouped launch + one reduction launch
    (tmjx_gemm_dw_grouped), stream-ordered behind the backward pass — nothing reads the views before that.  Layers whose operands are not
    16-byte aligned (the 1-wide value head) and parameters without a flat gradient view compute their gradient on the spot as usual."""
    active = None

    def __enter__(self):
        self.problems, self.keep, self.colsums = [], [], []
        # bf16 GEMM-input mode: the layers' weight gradients (networks.bgemm_dw) as one grouped launch too (tmjx_bgemm_dw_grouped; TMJX_BDW_GROUPED=0: layer by layer)
        self.bproblems, self.group_bf16 = [], os.environ.get("TMJX_BDW_GROUPED", "1") != "0"
        self.seen: set = set()       # ids of the parameters whose gradient views have been handed to autograd in this block
        deferred_weight_grads.active = self
        return self

    def __exit__(self, *a):
        deferred_weight_grads.active = None
        return False

    @staticmethod
    def _aligned(t):
        return t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1

    def try_add(self, dy2, x2, w, b):
        gw, gb = w.grad, (b.grad if b is not None else None)
        if gw is None or (b is not None and gb is None) or len(self.problems) >= 16:
            return None
        if not (self._aligned(dy2) and self._aligned(x2) and gw.stride(1) == 1 and gw.shape == w.shape):
            return None
        if id(w) in self.seen:
            # the same weight a second time in one graph (a shared layer, a network applied to two inputs): autograd is about to ADD this
            # use's gradient to the view it was handed for the first use — so that view must hold the first use's gradient by then.  Compute
            # the recorded problem of this weight NOW (stream-ordered in front of autograd's add) and give this use a fresh tensor
            mine = [q for q in self.problems if q[2].data_ptr() == gw.data_ptr()]
            self.problems = [q for q in self.problems if q[2].data_ptr() != gw.data_ptr()]
            self._launch_problems(mine)
            return None
        self.seen.add(id(w))
        self.problems.append((dy2, x2, gw, gb))
        return gw, gb

    def launch_subset(self, pick, target_wgs: int = 0):
        """The recorded problems whose gradient view `pick(gw)` selects, NOW (stream-ordered on the current stream), as a group of their own with
        a workgroup budget of `target_wgs` (0: the default); the others stay recorded for launch().  PPOLearner uses it to run the value network's
        weight gradients on the side stream right behind that network's backward pass, next to the policy's backward pass on the main stream."""
        mine = [q for q in self.problems if pick(q[2])]
        self.problems = [q for q in self.problems if not pick(q[2])]
        self._launch_problems(mine, target_wgs)

    def launch(self):
        import ctypes as C
        from .. import hip as _hip
        L = _hip.lib()
        if self.colsums:      # the LayerNorm blocks' (d gamma | d beta | d bias) partials (networks._dx_through_block)
            dev = self.colsums[0][0].device
            arr = (_hip.ColsumProblem * len(self.colsums))(*[_hip.ColsumProblem(pt.data_ptr(), g.data_ptr(), nb, wd) for pt, g, nb, wd in self.colsums])
            with torch.cuda.device(dev):
                _hip.check(L.tmjx_colsum_grouped(arr, len(self.colsums), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tmjx_colsum_grouped")
            self.keep += [t for c in self.colsums for t in c[:2]]
            self.colsums = []
        self._launch_problems(self.problems)
        self.problems = []
        self._launch_bproblems()

    def _launch_bproblems(self):
        import ctypes as C
        from .. import hip as _hip
        probs, self.bproblems = self.bproblems, []
        if not probs:
            return
        L = _hip.lib()
        dev = probs[0][0].device
        for at in range(0, len(probs), 24):
            grp = probs[at:at + 24]
            sizes = [int(L.tmjx_bgemm_dw_scratch_floats(dy.shape[0], dy.shape[1], x.shape[1])) for dy, x, _, _ in grp]
            scratch = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), dtype=torch.float32, device=dev)
            arr = (_hip.BdwProblem * len(grp))()
            off = 0
            for i, (dy, x, dw, db) in enumerate(grp):
                N, K = dy.shape[1], x.shape[1]
                arr[i] = _hip.BdwProblem(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, scratch.data_ptr() + 4 * off,
                                         int(dy.dtype == torch.float32), int(x.dtype == torch.float32), _ld(dy), _ld(x), dw.stride(0) if N > 1 else max(dw.stride(0), K),
                                         dy.shape[0], N, K)
                off += (sizes[i] + 3) // 4 * 4
            with torch.cuda.device(dev):
                _hip.check(L.tmjx_bgemm_dw_grouped(arr, len(grp), 0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tmjx_bgemm_dw_grouped")
            self.keep += [scratch] + [t for q in grp for t in q[:2]]

    def _launch_problems(self, problems, target_wgs: int = 0):
        import ctypes as C
        from .. import hip as _hip
        if not problems:
            return
        L = _hip.lib()
        dev = problems[0][0].device
        sizes = [int(L.tmjx_gemm_dw_scratch_floats(dy.shape[0], dy.shape[1], x.shape[1])) for dy, x, _, _ in problems]
        scratch = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), dtype=torch.float32, device=dev)
        arr = (_hip.DwProblem * len(problems))()
        off = 0
        for i, (dy, x, gw, gb) in enumerate(problems):
            arr[i] = _hip.DwProblem(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else None,
                                    scratch.data_ptr() + 4 * off, dy.stride(0), x.stride(0), gw.stride(0), dy.shape[0], dy.shape[1], x.shape[1])
            off += (sizes[i] + 3) // 4 * 4
        with torch.cuda.device(dev):
            _hip.check(L.tmjx_gemm_dw_grouped_wgs(arr, len(problems), int(target_wgs), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tmjx_gemm_dw_grouped")
        self.keep += [scratch] + [t for q in problems for t in q[:2]]


# ---- whole-chain fp32 kernels (csrc/mlp_chain.h): nets whose hidden layers are exactly 256 wide -----------------------------------------
def chain_fwd_desc(x2, hidden, final=None, kind="ln", eps=1e-6, out=None, latent=None):
    """The tmjx_chain_fwd_t of one forward chain and the tensors it writes.  x2: [M, >= K0] fp32 rows (a column prefix of a wider buffer is
    fine); hidden: [(weight [256, K], bias, gamma, beta)] ("ln") or [(weight, bias)] ("silu"); final: (weight [Nf, 256], bias) or None.
    Returns (desc, saved = [(z, y, stats)] per hidden layer, out [M, Nf] / [M] for a 1-wide head / None, keep-alive list)."""
    from .. import hip as _hip
    M, dev = x2.shape[0], x2.device
    d = _hip.ChainFwd()
    d.A, d.lda, d.M, d.n_hidden, d.epi, d.eps = x2.data_ptr(), x2.stride(0), M, len(hidden), 1 if kind == "ln" else 3, float(eps)
    saved = []
    d.rows_alloc = rows = int(_hip.lib().tmjx_chain_rows(M))       # (a y that feeds the next layer of the launch leaves the CU as whole row tiles: padding rows)
    for l, h in enumerate(hidden):
        w, b = h[0], h[1]
        z = torch.empty((M, 256), dtype=torch.float32, device=dev)
        y = torch.empty((rows, 256), dtype=torch.float32, device=dev)[:M]
        st = torch.empty((M, 2), dtype=torch.float32, device=dev) if kind == "ln" else None
        L = d.hidden[l]
        L.W, L.bias, L.z, L.y, L.K, L.ldw = w.data_ptr(), b.data_ptr(), z.data_ptr(), y.data_ptr(), w.shape[1], w.stride(0)
        if kind == "ln":
            L.gamma, L.beta, L.stats = h[2].data_ptr(), h[3].data_ptr(), st.data_ptr()
        saved.append((z, y, st))
    if final is not None:
        wf, bf = final
        Nf = wf.shape[0]
        if out is None:
            out = torch.empty((M,) if Nf == 1 else (M, Nf), dtype=torch.float32, device=dev)
        d.Wf, d.bf, d.outf, d.Nf, d.ldwf, d.ldof = wf.data_ptr(), (bf.data_ptr() if bf is not None else None), out.data_ptr(), Nf, wf.stride(0), (1 if Nf == 1 else out.stride(0))
    if latent is not None:
        # the encoder chain's latent tail: (eps [M, Z], dec_in [M, ld] to write, prop = the proprioceptive columns [M, W] as a view with its own row stride)
        leps, dec_in, prop = latent
        d.lat_eps, d.lat_out, d.prop = leps.data_ptr(), dec_in.data_ptr(), prop.data_ptr()
        d.lat_Z, d.lat_ld, d.prop_w, d.prop_ld = leps.shape[1], dec_in.stride(0), prop.shape[1], prop.stride(0)
    return d, saved, out


def chain_fwd(x2, hidden, final=None, kind="ln", eps=1e-6, latent=None):
    """One launch: the whole chain forward (tmjx_chain_fwd).  Raises TmjxError (EINVAL) when the chain does not qualify: ask chain_fwd_ok first."""
    import ctypes as C
    d, saved, out = chain_fwd_desc(x2, hidden, final, kind, eps, latent=latent)
    _launch("tmjx_chain_fwd", x2.device, C.byref(d))
    return saved, out


def chain_fwd_ok(x2, hidden, final=None, kind="ln") -> bool:
    import ctypes as C
    from .. import hip as _hip
    if os.environ.get("TMJX_NO_CHAIN") or not (x2.is_cuda and x2.dtype == torch.float32 and x2.dim() == 2 and x2.stride(1) == 1):
        return False
    if any(h[0].shape[0] != 256 for h in hidden) or not (1 <= len(hidden) <= 4):
        return False
    return bool(_hip.lib().tmjx_chain_fwd_ok(C.byref(chain_fwd_desc(x2[:1], hidden, final, kind)[0])))


def chain_bwd(g, final_w, blocks, kind="ln", w0=None, dx_cols=None, prof=None, dx_ld=None):
    """One launch: the backward chain (tmjx_chain_bwd).  g: d loss / d (last layer's output) [M, Kg] (a 1-wide head: [M]); final_w: the last layer's
    weight [Kg, 256]; blocks: the hidden layers LAST FIRST as (weight [256, K], z, bias, gamma, stats) ("ln") or (weight, z, bias) ("silu") — the
    weight of blocks[i] is the operand of stage i + 1; w0 / dx_cols: the first hidden layer's weight and how many of its input columns need a gradient.
    Returns (dz per block (last first), partials per block or None, dx [M, ceil4(dx_cols)] or None)."""
    import ctypes as C
    from .. import hip as _hip
    L = _hip.lib()
    M, dev = g.shape[0], g.device
    d = _hip.ChainBwd()
    head = g.dim() == 1
    d.G, d.ldg, d.Kg, d.M, d.n_stages, d.epi = g.data_ptr(), (1 if head else g.stride(0)), (1 if head else g.shape[1]), M, len(blocks), 2 if kind == "ln" else 4
    dzs, partials = [], []
    nfl = int(L.tmjx_gemm_nn_ln_bwd_partial_floats(M, 256)) if kind == "ln" else 0
    d.rows_alloc = rows = int(L.tmjx_chain_rows(M))
    for i, blk in enumerate(blocks):
        S = d.stage[i]
        w = final_w if i == 0 else blocks[i - 1][0]
        S.W, S.ldw = w.data_ptr(), (w.stride(0) if w.dim() == 2 and w.shape[0] > 1 else 256)
        dz = torch.empty((rows, 256), dtype=torch.float32, device=dev)[:M]
        S.z, S.bias, S.dz = blk[1].data_ptr(), blk[2].data_ptr(), dz.data_ptr()
        if kind == "ln":
            pt = torch.empty(nfl, dtype=torch.float32, device=dev)
            S.gamma, S.stats, S.partial = blk[3].data_ptr(), blk[4].data_ptr(), pt.data_ptr()
            partials.append(pt)
        dzs.append(dz)
    dx = None
    if w0 is not None:
        cols = int(dx_cols)
        # (dx_ld: a buffer of the input's own width, of which only the first `cols` columns are written — what autograd wants back for the input)
        dx = torch.empty((M, (cols + 3) // 4 * 4 if dx_ld is None else int(dx_ld)), dtype=torch.float32, device=dev)
        d.W0, d.ldw0, d.dx_cols, d.dx, d.lddx = w0.data_ptr(), w0.stride(0), cols, dx.data_ptr(), dx.stride(0)
    if prof is not None:
        d.prof = prof.data_ptr()
    _launch("tmjx_chain_bwd", dev, C.byref(d))
    return dzs, (partials if kind == "ln" else None), dx


# ---- a whole MLP chain as ONE autograd function (bf16 GEMM-input mode) -------------------------------------------------------------------
class _Layer:
    """One layer of a chain: kind "ln" (Dense -> SiLU -> LayerNorm block), "silu" (Dense -> SiLU, brax value MLP), "dense" (un-activated) or
    "head" (the 1-wide un-activated last layer of the value MLP, on the fp32 kernels; behind a "silu" layer)."""
    __slots__ = ("kind", "lin", "norm", "fused")

    def __init__(self, kind, lin, norm=None):
        from .. import hip as _hip
        self.kind, self.lin, self.norm = kind, lin, norm
        # whole-row epilogues need the layer to be one tile wide; Dense -> SiLU epilogues are element-wise (any width)
        self.fused = kind == "silu" or (kind == "ln" and bool(_hip.lib().tmjx_bgemm_row_tile_ok(lin.out_features)))

    def params(self):
        return [self.lin.weight, self.lin.bias] + ([self.norm.weight, self.norm.bias] if self.kind == "ln" else [])


class _BfChainFn(torch.autograd.Function):
    """y = chain(x) for a sequence of layers with every contraction on the bf16 kernels (csrc/gemm_bf16.h) and every block epilogue fused:
    forward one launch per layer (tmjx_bgemm_ln_fwd / _silu_fwd / _nt), the hidden activations exist only as bf16 (the next GEMM's operand),
    and so do the pre-activations z saved for the backward pass (the 1024-wide block's stay fp32); backward per layer ONE launch for the input gradient with the PRODUCING block's
    LayerNorm / SiLU backward in its epilogue (tmjx_bgemm_ln_bwd / _silu_bwd: d loss / d z as bf16 + column-sum partials) and one for the
    weight gradient (tmjx_bgemm_dw, transposed LDS reads), written straight into the flat gradient buffer's views where the learner provides
    them; all column-sum partials are reduced by one grouped launch at the end.  Layers that are not one tile wide (the 1024-wide first
    encoder block) take the unfused kernels (tmjx_bgemm_nt + tmjx_silu_ln_fwd_bf16 / _bwd_bf16: fp32 z, bf16 results).  `last_y_f32`: the chain
    ends with a block whose output goes to an fp32 kernel (the value net's last hidden layer in front of the 1-wide head, which runs inside the chain:
    kind "head").  A first layer whose fp32 input has a registered bf16 twin (gemm_inputs.twins: the minibatch's normalised observations) stages the twin."""

    @staticmethod
    def forward(ctx, x, layers, sh, dx_cols, last_y_f32, *params):
        import ctypes as C
        from .. import hip as _hip
        x2 = _rows2d(x)
        q = 4 if x2.dtype == torch.float32 else 8
        if x2.data_ptr() % 16 or x2.stride(0) % q or x2.stride(1) != 1:
            # rows that do not start on 16-byte boundaries (a torch.cat of 286 columns: the policy's small-batch path): a copy with padded rows
            padded = torch.empty((x2.shape[0], (x2.shape[1] + q - 1) // q * q), dtype=x2.dtype, device=x2.device)
            padded[:, :x2.shape[1]].copy_(x2)
            x2 = padded[:, :x2.shape[1]]
        saved, h = [], x2
        t16 = gemm_inputs.twin_of(x2, layers[0].lin.weight.shape[1])
        for i, L in enumerate(layers):
            lin = L.lin
            N, K = lin.weight.shape
            last = i == len(layers) - 1 or layers[i + 1].kind == "head"
            hx = h                 # the layer's input as the weight gradient wants it (true width)
            if i == 0 and t16 is not None:
                # columns K .. ceil64(K) of the twin (other observation columns, or its zero padding) meet the shadow's zero columns
                h, hx, K = t16, t16[:, :K], t16.shape[1]
            if L.kind == "dense":
                y = bgemm_nt(h, sh.w[lin], N, K, lin.bias)
                saved.append((hx, None, None))
            elif L.kind == "head":
                y = gemm_nt(h, lin.weight, lin.bias)          # fp32 kernel on the fp32 activation of the last hidden layer (last_y_f32)
                saved.append((h, None, None))
            elif L.kind == "silu":
                z, y = bgemm_silu_fwd(h, sh.w[lin], N, K, lin.bias, y_f32=last and last_y_f32)
                saved.append((hx, z, None))
            elif L.fused:
                z, y, stats = bgemm_ln_fwd(h, sh.w[lin], N, K, lin.bias, L.norm.weight, L.norm.bias, L.norm.eps)
                saved.append((hx, z, stats))
            else:
                # not one tile wide: plain GEMM + the row kernel, whose result goes out as bf16 — what the next GEMM would make of it anyway
                z = bgemm_nt(h, sh.w[lin], N, K)
                stats = torch.empty((z.shape[0], 2), dtype=torch.float32, device=z.device)
                if last:
                    y = torch.empty_like(z)
                    _launch("tmjx_silu_ln_fwd", z.device, _p(z), _p(lin.bias), _p(L.norm.weight), _p(L.norm.bias), _p(y), _p(stats), z.shape[0], N, float(L.norm.eps))
                else:
                    y = torch.empty((z.shape[0], N), dtype=torch.bfloat16, device=z.device)
                    _launch("tmjx_silu_ln_fwd_bf16", z.device, _p(z), _p(lin.bias), _p(L.norm.weight), _p(L.norm.bias), _p(y), N, _p(stats), z.shape[0], N, float(L.norm.eps))
                saved.append((hx, z, stats))
            h = y
        ctx.layers, ctx.sh, ctx.dx_cols, ctx.saved, ctx.x_shape = layers, sh, dx_cols, saved, x.shape
        return h.view(*x.shape[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dout):
        from .. import hip as _hip
        layers, sh, saved = ctx.layers, ctx.sh, ctx.saved
        Lh = _hip.lib()
        d = deferred_weight_grads.active
        g = _rows2d(dout)
        if g.data_ptr() % 16 or g.stride(0) % 4:
            g = g.contiguous()
        M = g.shape[0]
        g_is_dz = layers[-1].kind in ("dense", "head")
        grads = {}                 # id(param) -> gradient tensor
        colsums = []               # (partial, out, rows, width): reduced by one grouped launch

        def dest(p):               # the flat-buffer view to write a weight gradient into, if the learner provides one
            if d is not None and p.grad is not None and p.grad.shape == p.shape and p.grad.stride(-1) == 1 and id(p) not in d.seen:
                d.seen.add(id(p))
                return p.grad
            return None

        def block_sums(L, partial):
            N = L.lin.out_features
            if L.kind == "ln":
                out = torch.empty((3, N), dtype=torch.float32, device=partial.device)
                colsums.append((partial, out, partial.shape[0], 3 * N))
                grads[id(L.norm.weight)], grads[id(L.norm.bias)], grads[id(L.lin.bias)] = out[0], out[1], out[2]
            else:
                out = torch.empty(N, dtype=torch.float32, device=partial.device)
                colsums.append((partial, out, partial.shape[0], N))
                grads[id(L.lin.bias)] = out

        dx = None
        for i in range(len(layers) - 1, -1, -1):
            L = layers[i]
            lin = L.lin
            N, K = lin.weight.shape
            h, z, stats = saved[i]
            if not g_is_dz:        # g = d loss / d y of this block: its own (unfused) backward
                if L.kind == "ln":
                    gy = g if g.dtype == torch.float32 and g.is_contiguous() else g.float().contiguous()
                    # d loss / d z as bf16: its two consumers are bf16-operand GEMMs (this layer's weight gradient, the producer's input gradient)
                    dz = torch.empty((M, N), dtype=torch.bfloat16, device=z.device)
                    g3 = torch.empty((3, N), dtype=torch.float32, device=z.device)
                    partial = torch.empty(int(Lh.tmjx_silu_ln_partial_floats(M, N)), dtype=torch.float32, device=z.device)
                    # (the row kernel reads z as fp32; a FUSED ln block saves its z as bf16 — a chain that ends in one takes the upcast copy)
                    z = z if z.dtype == torch.float32 else z.float()
                    _launch("tmjx_silu_ln_bwd_bf16", z.device, _p(gy), _p(z), _p(lin.bias), _p(L.norm.weight), _p(stats), _p(dz), N, _p(g3), _p(partial), M, N)
                    grads[id(L.norm.weight)], grads[id(L.norm.bias)], grads[id(lin.bias)] = g3[0], g3[1], g3[2]
                    g = dz
                else:
                    gy = g if g.dtype == torch.float32 else g.float()
                    ld = (N + 7) // 8 * 8
                    dz = torch.empty((M, ld), dtype=torch.bfloat16, device=z.device)
                    partial = torch.empty(int(Lh.tmjx_bgemm_partial_floats(M, N, 1)), dtype=torch.float32, device=z.device)
                    _launch("tmjx_bf_silu_bwd", z.device, _p(gy), gy.stride(0), _p(z), z.stride(0), _p(lin.bias), _p(dz), ld, _p(partial), M, N)
                    block_sums(L, partial.view(-1, N))
                    g = dz[:, :N]
            # g = d loss / d z of layer i
            with_bias = L.kind in ("dense", "head")
            if L.kind == "head":
                dw, db = gemm_dw(g, h, True)                  # fp32 kernels, as the stand-alone head layer's backward (_HipDenseFn)
            else:
                dw, db = bgemm_dw(g, h, with_bias, out=dest(lin.weight), out_bias=dest(lin.bias) if with_bias else None)
            grads[id(lin.weight)] = dw
            if with_bias:
                grads[id(lin.bias)] = db
            if i > 0:
                P = layers[i - 1]
                Np = P.lin.out_features
                ph, pz, pstats = saved[i - 1]
                if L.kind == "head":
                    # d loss / d y of the last hidden layer is the outer product g[:, 0] x (the head's weight row): formed inside that layer's SiLU
                    # backward (tmjx_bf_silu_bwd_rank1) instead of by an input-gradient GEMM with a contraction length of one
                    ld = (Np + 7) // 8 * 8
                    dzp = torch.empty((M, ld), dtype=torch.bfloat16, device=g.device)
                    partial = torch.empty(int(Lh.tmjx_bgemm_partial_floats(M, Np, 1)), dtype=torch.float32, device=g.device)
                    g1 = g.reshape(-1)
                    _launch("tmjx_bf_silu_bwd_rank1", g.device, _p(g1 if g1.is_contiguous() else g1.contiguous()), _p(lin.weight), _p(pz), pz.stride(0), _p(P.lin.bias), _p(dzp), ld,
                            _p(partial), M, Np)
                    block_sums(P, partial.view(-1, Np))
                    g, g_is_dz = dzp[:, :Np], True
                elif P.fused and P.kind == "ln":
                    g, partial = bgemm_ln_bwd(g, sh.wt[lin], Np, N, pz, P.lin.bias, P.norm.weight, pstats)
                    block_sums(P, partial)
                    g_is_dz = True
                elif P.fused:
                    g, partial = bgemm_silu_bwd(g, sh.wt[lin], Np, N, pz, P.lin.bias)
                    block_sums(P, partial)
                    g_is_dz = True
                else:
                    g = bgemm_nt(g, sh.wt[lin], Np, N)
                    g_is_dz = False
            elif ctx.needs_input_grad[0]:
                cols = K if ctx.dx_cols is None else int(ctx.dx_cols)
                dx = torch.empty((M, K), dtype=torch.float32, device=g.device)
                bgemm_nt(g, sh.wt[lin], cols, N, out=dx)
                dx = dx.view(ctx.x_shape)
        if colsums:
            arr = (_hip.ColsumProblem * len(colsums))(*[_hip.ColsumProblem(pt.data_ptr(), o.data_ptr(), nb, wd) for pt, o, nb, wd in colsums])
            _launch("tmjx_colsum_grouped", colsums[0][0].device, arr, len(colsums))
            ctx._keep = colsums
        out = []
        for L in layers:
            out += [grads.get(id(p)) for p in L.params()]
        return (dx, None, None, None, None, *out)


def bf16_chain(x, layers, dx_cols=None, last_y_f32=False):
    params = [p for L in layers for p in L.params()]
    return _BfChainFn.apply(x, layers, gemm_inputs.shadows, dx_cols, last_y_f32, *params)


def _bf16_chain_ok(x) -> bool:
    sh = gemm_inputs.shadows
    return gemm_inputs.dtype == torch.bfloat16 and sh is not None and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not os.environ.get("TMJX_NO_BF16_CHAIN")



class _F32ChainFn(torch.autograd.Function):
    """y = chain(x) for an fp32 chain whose hidden layers are all 256 wide, forward AND backward ONE launch each (tmjx_chain_fwd / tmjx_chain_bwd,
    csrc/mlp_chain.h): the encoder's blocks + fc2, the decoder's blocks + the action head (intention_network.py:32-44,68-76,128-139), brax's value MLP
    with its 1-wide head (ppo_networks.py:180-184).  Bit-identical to the layer-by-layer functions (_HipBlockFn / _HipDenseFn / _ValueChainFn): same
    saved tensors, same expressions; weight gradients go to the learner's grouped launch (deferred_weight_grads) as before, the LayerNorm blocks'
    (d gamma | d beta | d bias) column-sum partials to its grouped reduction, the 1-wide head's gradients through tmjx_head_dw.  `layers`: _Layer list,
    hidden layers ("ln" or "silu", all of one kind) then the last layer ("dense" / "head"); `dx_cols`: the input's leading columns that need a gradient."""

    @staticmethod
    def forward(ctx, x, layers, dx_cols, latent, *params):
        x2 = _rows2d(x)
        hid, last = layers[:-1], layers[-1]
        kind = hid[0].kind
        hidden = [(L.lin.weight, L.lin.bias, L.norm.weight, L.norm.bias) if kind == "ln" else (L.lin.weight, L.lin.bias) for L in hid]
        eps = float(hid[0].norm.eps) if kind == "ln" else 0.0
        dec_in = None
        if latent is not None:
            # the encoder: the launch also writes the decoder's input [latent sample | proprioception] (the kernel's latent tail); handed out as a second,
            # non-differentiable output — _LatentViewFn ties it to fc2 for the backward pass
            leps, prop = latent
            wd = leps.shape[1] + prop.shape[1]
            dec_in = torch.empty((x2.shape[0], (wd + 3) // 4 * 4), dtype=torch.float32, device=x2.device)
            latent = (leps, dec_in, prop)
        saved, out = chain_fwd(x2, hidden, (last.lin.weight, last.lin.bias), kind, eps, latent=latent)
        flat = [t for z, y, st in saved for t in ((z, y, st) if kind == "ln" else (z, y))]
        ctx.save_for_backward(x2, *flat, *params)
        ctx.layers, ctx.dx_cols, ctx.x_shape, ctx.kind, ctx.nflat = layers, dx_cols, x.shape, kind, len(flat)
        res = out.view(*x.shape[:-1], last.lin.out_features) if last.kind == "dense" else out.view(*x.shape[:-1], 1)
        if dec_in is None:
            return res
        xd = dec_in[:, :wd]
        ctx.mark_non_differentiable(xd)
        return res, xd

    @staticmethod
    def backward(ctx, dout, *_unused):
        from .. import hip as _hip
        layers, kind = ctx.layers, ctx.kind
        hid, last = layers[:-1], layers[-1]
        st_ = ctx.saved_tensors
        x2, flat, params = st_[0], st_[1:1 + ctx.nflat], st_[1 + ctx.nflat:]
        per = 3 if kind == "ln" else 2
        saved = [flat[per * i:per * i + per] for i in range(len(hid))]
        # the parameters as they were at forward time (saved tensors; version-checked by autograd), in _Layer.params() order
        pit = iter(params)
        P = [[next(pit) for _ in L.params()] for L in layers]
        Lh = _hip.lib()
        dev = dout.device
        M = x2.shape[0]
        d = deferred_weight_grads.active
        grads: dict = {}
        wf, bf = P[-1][0], P[-1][1]
        y_last = saved[-1][1]
        if last.kind == "head":
            g = dout.reshape(-1)
            g = g if g.is_contiguous() else g.contiguous()
            K = wf.shape[1]
            dwh, dbh = torch.empty((1, K), dtype=torch.float32, device=dev), torch.empty(1, dtype=torch.float32, device=dev)
            scratch = torch.empty(int(Lh.tmjx_head_dw_scratch_floats(M, K)), dtype=torch.float32, device=dev)
            _launch("tmjx_head_dw", dev, _p(g), _p(y_last), y_last.stride(0), _p(dwh), _p(dbh), _p(scratch), M, K)
            grads[(len(layers) - 1, 0)], grads[(len(layers) - 1, 1)] = dwh, dbh
            ctx._keep = scratch
        else:
            g = _rows2d(dout)
            if g.data_ptr() % 16 or g.stride(0) % 4:
                g = g.contiguous()
            got = d.try_add(g, y_last, last.lin.weight, last.lin.bias) if d is not None else None
            dw, db = got if got is not None else gemm_dw(g, y_last, True)
            grads[(len(layers) - 1, 0)], grads[(len(layers) - 1, 1)] = dw, db
        if kind == "ln":
            blocks = [(P[l][0], saved[l][0], P[l][1], P[l][2], saved[l][2]) for l in range(len(hid) - 1, -1, -1)]
        else:
            blocks = [(P[l][0], saved[l][0], P[l][1]) for l in range(len(hid) - 1, -1, -1)]
        want_dx = ctx.needs_input_grad[0]
        cols = (x2.shape[1] if ctx.dx_cols is None else int(ctx.dx_cols)) if want_dx else None
        dzs, partials, dx = chain_bwd(g, wf, blocks, kind, P[0][0] if want_dx else None, cols, dx_ld=x2.shape[1] if want_dx else None)
        for i, dz in enumerate(dzs):
            l = len(hid) - 1 - i
            L = hid[l]
            xin = x2 if l == 0 else saved[l - 1][1]
            if kind == "ln":
                got = d.try_add(dz, xin, L.lin.weight, None) if d is not None else None
                grads[(l, 0)] = got[0] if got is not None else gemm_dw(dz, xin, False)[0]
                g3 = torch.empty((3, 256), dtype=torch.float32, device=dev)
                nblk = partials[i].numel() // 768
                if d is not None and len(d.colsums) < 16:
                    d.colsums.append((partials[i], g3, nblk, 768))      # reduced with the others in ONE launch behind the backward pass (launch())
                else:
                    one = (_hip.ColsumProblem * 1)(_hip.ColsumProblem(partials[i].data_ptr(), g3.data_ptr(), nblk, 768))
                    _launch("tmjx_colsum_grouped", dev, one, 1)
                grads[(l, 2)], grads[(l, 3)], grads[(l, 1)] = g3[0], g3[1], g3[2]
            else:
                got = d.try_add(dz, xin, L.lin.weight, L.lin.bias) if d is not None else None
                grads[(l, 0)], grads[(l, 1)] = got if got is not None else gemm_dw(dz, xin, True)
        out = [grads.get((l, j)) for l, L in enumerate(layers) for j in range(len(L.params()))]
        return (dx.view(ctx.x_shape) if dx is not None else None, None, None, None, *out)


def f32_chain(x, layers, dx_cols=None, latent=None):
    params = [p for L in layers for p in L.params()]
    return _F32ChainFn.apply(x, layers, dx_cols, latent, *params)


def _f32_chain_ok(x, layers, dx_cols=None) -> bool:
    """The whole-chain fp32 kernels take it: fp32 mode on the GPU, every hidden layer 256 wide, a last layer of at most 128 columns, aligned rows,
    and the chain's input gradient (if any) limited to at most 128 leading columns."""
    if gemm_inputs.dtype is not None or torch.is_autocast_enabled() or os.environ.get("TMJX_NO_CHAIN") or not (x.is_cuda and x.dtype == torch.float32):
        return False
    if x.requires_grad and torch.is_grad_enabled() and (x.shape[-1] if dx_cols is None else int(dx_cols)) > 128:
        return False
    hid, last = layers[:-1], layers[-1]
    if not hid or any(L.lin.out_features != 256 or L.lin.bias is None for L in hid) or last.lin.bias is None or last.lin.in_features != 256:
        return False
    kind = hid[0].kind
    hidden = [(L.lin.weight, L.lin.bias, L.norm.weight, L.norm.bias) if kind == "ln" else (L.lin.weight, L.lin.bias) for L in hid]
    return chain_fwd_ok(_rows2d(x), hidden, (last.lin.weight, last.lin.bias), kind)


class _BlockLink:
    """Hand-over between a Dense -> SiLU -> LayerNorm block and the ONE layer that consumes its output (inside `ln_bwd_links()`): the consumer's
    backward applies the block's LayerNorm + SiLU backward in the epilogue of its own input-gradient GEMM (tmjx_gemm_nn_ln_bwd) and hands the
    block d loss / d z instead of d loss / d y, plus the block's (d gamma, d beta, d bias)."""
    __slots__ = ("y_ptr", "y_shape", "ctx", "dz_given", "grads")

    def __init__(self):
        self.y_ptr, self.y_shape, self.ctx, self.dz_given, self.grads = None, None, None, False, None


class ln_bwd_links:
    """`with ln_bwd_links():` around a forward pass whose blocks form chains (each block's output feeds exactly one dense layer, as in the
    intention network's encoder and decoder): enables the fused LayerNorm-backward epilogue.  Outside it every block runs its own backward."""
    active = False
    last = None      # link of the most recent block's output

    def __enter__(self):
        self._prev = (ln_bwd_links.active, ln_bwd_links.last)
        ln_bwd_links.active, ln_bwd_links.last = not os.environ.get("TMJX_NO_LN_BWD_FUSION"), None
        return self

    def __exit__(self, *a):
        ln_bwd_links.active, ln_bwd_links.last = self._prev
        return False

    @staticmethod
    def producer_of(x2):
        """The link of the block whose output IS x2 (same storage, same shape), or None."""
        l = ln_bwd_links.last
        if ln_bwd_links.active and l is not None and l.y_ptr == x2.data_ptr() and l.y_shape == tuple(x2.shape) and x2.is_contiguous():
            return l
        return None


def _dx_through_block(dy2, w, link):
    """d loss / d z of the producing block from this layer's output gradient dy2 [M, N] and weight w [N, 256]: one launch."""
    import ctypes as C
    from .. import hip as _hip
    L = _hip.lib()
    x2, wp, z, b, gamma, stats = link.ctx.saved_tensors
    M, H = z.shape
    dz = torch.empty_like(z)
    nfl = int(L.tmjx_gemm_nn_ln_bwd_partial_floats(M, H))
    nblk = nfl // (3 * H)          # one partial row per workgroup of the launch: 80- or 32-row tiles by M (csrc/tmjx_hip.hip: gemm_mt)
    partial = torch.empty(nfl, dtype=torch.float32, device=z.device)
    grads = torch.empty((3, H), dtype=torch.float32, device=z.device)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(z.device):
        stream = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        _hip.check(L.tmjx_gemm_nn_ln_bwd(p(dy2), dy2.stride(0), p(w), w.stride(0), p(z), p(b), p(gamma), p(stats), p(dz), p(partial), M, H, dy2.shape[1], stream),
                   "tmjx_gemm_nn_ln_bwd")
        d = deferred_weight_grads.active
        if d is not None and len(d.colsums) < 16:
            d.colsums.append((partial, grads, nblk, 3 * H))      # reduced with the others in ONE launch behind the backward pass (launch())
        else:
            one = (_hip.ColsumProblem * 1)(_hip.ColsumProblem(partial.data_ptr(), grads.data_ptr(), nblk, 3 * H))
            _hip.check(L.tmjx_colsum_grouped(one, 1, stream), "tmjx_colsum_grouped")
    link.dz_given, link.grads = True, grads
    return dz


def _fusable_dx(dy2, w, link, dx_cols) -> bool:
    if link is None or dx_cols is not None or link.ctx is None:
        return False
    import ctypes as C
    from .. import hip as _hip
    z = link.ctx.saved_tensors[2]
    return bool(z.shape[1] == w.shape[1] and _hip.lib().tmjx_gemm_nn_ln_bwd_ok(C.c_void_p(dy2.data_ptr()), dy2.stride(0), C.c_void_p(w.data_ptr()), w.stride(0), z.shape[1]))


class _HipDenseFn(torch.autograd.Function):
    """y = x W^T (+ b) with all three contractions on the library's MFMA kernels: forward tmjx_gemm_nt, input gradient tmjx_gemm_nn,
    weight + bias gradient tmjx_gemm_dw.  `dx_cols`: the caller only needs the gradient of the first dx_cols input columns."""

    @staticmethod
    def forward(ctx, x, w, b, dx_cols):
        x2 = _rows2d(x)
        ctx.save_for_backward(x2, w)
        ctx.has_bias, ctx.dx_cols, ctx.x_shape = b is not None, dx_cols, x.shape
        ctx.params = (w, b)                    # the Parameter objects (their .grad = the flat-buffer views a deferred gradient lands in)
        ctx.producer = ln_bwd_links.producer_of(x2)
        ln_bwd_links.last = None
        return gemm_nt(x2, w, b).view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = _rows2d(dy)
        if ctx.needs_input_grad[0] and _fusable_dx(dy2, w, ctx.producer, ctx.dx_cols):
            dx = _dx_through_block(dy2, w, ctx.producer).view(ctx.x_shape)      # (d loss / d z of the producing block: its backward skips its own kernel)
        else:
            dx = gemm_nn(dy2, w, ctx.dx_cols).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        d = deferred_weight_grads.active
        got = d.try_add(dy2, x2, *ctx.params) if d is not None else None
        dw, db = got if got is not None else gemm_dw(dy2, x2, ctx.has_bias)
        return dx, dw, db, None


class _HipSiluDenseFn(torch.autograd.Function):
    """y = silu(x W^T + b) (a hidden layer of brax's value MLP, ppo_networks.py:180-184), fp32: forward ONE launch (tmjx_gemm_nt_silu: the activation
    on the accumulators of the MFMA tile; z without the bias is kept for the backward pass), backward dz = dy silu'(z + b) (tmjx_silu_bwd), then the
    input gradient (tmjx_gemm_nn) and the weight + bias gradient (deferred into the grouped launch, or tmjx_gemm_dw) — no torch element-wise kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        import ctypes as C
        from .. import hip as _hip
        x2 = _rows2d(x)
        M, K = x2.shape
        N = w.shape[0]
        z = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        y = torch.empty_like(z)
        L = _hip.lib()
        if L.tmjx_gemm_nt_silu_ok(_p(x2), x2.stride(0), _p(w), w.stride(0)):
            _launch("tmjx_gemm_nt_silu", x2.device, _p(x2), x2.stride(0), _p(w), w.stride(0), _p(b), _p(z), _p(y), N, M, N, K)
        else:
            _launch("tmjx_gemm_nt", x2.device, _p(x2), x2.stride(0), _p(w), w.stride(0), None, _p(z), N, M, N, K)
            _launch("tmjx_silu_fwd", x2.device, _p(z), _p(b), _p(y), M, N)
        ctx.save_for_backward(x2, w, z, b)
        ctx.params, ctx.x_shape = (w, b), x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, z, b = ctx.saved_tensors
        M, N = z.shape
        dy2 = _rows2d(dy).contiguous()
        dz = torch.empty_like(z)
        _launch("tmjx_silu_bwd", z.device, _p(dy2), _p(z), _p(b), _p(dz), M, N)
        dx = gemm_nn(dz, w).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        d = deferred_weight_grads.active
        got = d.try_add(dz, x2, *ctx.params) if d is not None else None
        dw, db = got if got is not None else gemm_dw(dz, x2, True)
        return dx, dw, db


class _ValueChainFn(torch.autograd.Function):
    """The whole fp32 value MLP (brax make_value_network, ppo_networks.py:180-184: Dense -> swish ... Dense(1)) as ONE autograd function: forward as the per-layer
    functions do it (tmjx_gemm_nt_silu per hidden layer, tmjx_gemm_nt for the 1-wide head); backward WITHOUT an element-wise launch between the GEMMs and without a
    GEMM for the head: tmjx_head_dw (the head's gradients: a matrix-vector product), tmjx_silu_bwd_rank1 (the last hidden layer's d loss / d z from the head's
    outer-product input gradient), then per hidden layer tmjx_gemm_nn_silu_bwd (its input gradient with the PRODUCING layer's SiLU backward on the accumulators);
    weight gradients of the hidden layers into the learner's grouped launch as before.  Same expressions as tmjx_silu_bwd / tmjx_gemm_nn: the hidden layers'
    gradients keep their bits; the head's weight gradient is summed in another order than tmjx_gemm_dw's slabs.  (Critics whose hidden layers are all 256 wide
    take _F32ChainFn — one launch each way — instead.)  Activations, pre-activations and the parameters as they were at forward time travel through
    ctx.save_for_backward (version-checked, visible to saved-tensor hooks); `lins` only names the Parameter objects whose .grad views the deferred
    weight gradients land in."""

    @staticmethod
    def forward(ctx, x, lins, *params):
        x2 = _rows2d(x)
        M = x2.shape[0]
        h, flat = x2, []
        for lin in lins[:-1]:
            N, K = lin.weight.shape
            z = torch.empty((M, N), dtype=torch.float32, device=x2.device)
            y = torch.empty_like(z)
            _launch("tmjx_gemm_nt_silu", x2.device, _p(h), h.stride(0), _p(lin.weight), lin.weight.stride(0), _p(lin.bias), _p(z), _p(y), N, M, N, K)
            flat += [h, z]
            h = y
        head = lins[-1]
        out = gemm_nt(h, head.weight, head.bias)
        ctx.save_for_backward(*flat, h, *params)
        ctx.lins, ctx.x_shape = lins, x.shape
        return out.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dout):
        from .. import hip as _hip
        lins = ctx.lins
        nl = len(lins)
        st = ctx.saved_tensors
        saved = [(st[2 * i], st[2 * i + 1]) for i in range(nl - 1)]
        h_last = st[2 * (nl - 1)]
        W = [(st[2 * (nl - 1) + 1 + 2 * i], st[2 * (nl - 1) + 2 + 2 * i]) for i in range(nl)]      # (weight, bias) as saved at forward time
        L = _hip.lib()
        dev = dout.device
        dy1 = dout.reshape(-1)
        dy1 = dy1 if dy1.is_contiguous() else dy1.contiguous()
        M = dy1.shape[0]
        head = lins[-1]
        grads = {}
        # the head: gradients as a matrix-vector product, its input gradient never materialised
        K = head.in_features
        dwh, dbh = torch.empty((1, K), dtype=torch.float32, device=dev), torch.empty(1, dtype=torch.float32, device=dev)
        scratch = torch.empty(int(L.tmjx_head_dw_scratch_floats(M, K)), dtype=torch.float32, device=dev)
        _launch("tmjx_head_dw", dev, _p(dy1), _p(h_last), h_last.stride(0), _p(dwh), _p(dbh), _p(scratch), M, K)
        grads[id(head.weight)], grads[id(head.bias)] = dwh, dbh
        last = lins[-2]
        dz = torch.empty_like(saved[-1][1])
        _launch("tmjx_silu_bwd_rank1", dev, _p(dy1), _p(W[-1][0]), _p(saved[-1][1]), _p(W[-2][1]), _p(dz), M, last.out_features)
        d = deferred_weight_grads.active
        dx = None
        for i in range(nl - 2, -1, -1):
            lin = lins[i]
            w_i = W[i][0]
            xin, _ = saved[i]
            got = d.try_add(dz, xin, lin.weight, lin.bias) if d is not None else None
            dw, db = got if got is not None else gemm_dw(dz, xin, True)
            grads[id(lin.weight)], grads[id(lin.bias)] = dw, db
            if i > 0:
                prev, pz = lins[i - 1], saved[i - 1][1]
                dzp = torch.empty_like(pz)
                _launch("tmjx_gemm_nn_silu_bwd", dev, _p(dz), dz.stride(0), _p(w_i), w_i.stride(0), _p(pz), _p(W[i - 1][1]), _p(dzp), M, prev.out_features, lin.out_features)
                dz = dzp
            elif ctx.needs_input_grad[0]:
                dx = gemm_nn(dz, w_i).view(ctx.x_shape)
        ctx._keep = scratch
        out = []
        for lin in lins:
            out += [grads[id(lin.weight)], grads[id(lin.bias)]]
        return (dx, None, *out)


def _value_chain_ok(x2, lins) -> bool:
    """The fused fp32 value chain's preconditions: a 1-wide head behind at least one hidden layer, widths in fours, 16-byte aligned rows everywhere."""
    from .. import hip as _hip
    if os.environ.get("TMJX_VALUE_CHAIN", "1") == "0" or len(lins) < 2 or lins[-1].out_features != 1 or lins[-1].bias is None:
        return False
    L = _hip.lib()
    if x2.data_ptr() % 16 or x2.stride(0) % 4 or x2.stride(1) != 1:
        return False
    for lin in lins[:-1]:
        if lin.bias is None or lin.out_features % 4 or lin.in_features % 4 or lin.weight.data_ptr() % 16 or lin.weight.stride(0) % 4 or lin.bias.data_ptr() % 16:
            return False
    if not (lins[-1].in_features % 4 == 0 and lins[-1].weight.data_ptr() % 16 == 0 and bool(L.tmjx_gemm_nt_silu_ok(_p(x2), x2.stride(0), _p(lins[0].weight), lins[0].weight.stride(0)))):
        return False
    # the backward pass's fused input-gradient launches (tmjx_gemm_nn_silu_bwd: dY = a fresh dense [M][out] gradient — aligned whenever `out` is a multiple
    # of four —, W = that layer's weight): asked here, so that a weight view with an unaligned stride takes the per-layer path instead of failing in backward
    return all(bool(L.tmjx_gemm_nn_silu_bwd_ok(None, lin.out_features, _p(lin.weight), lin.weight.stride(0))) for lin in lins[1:-1])


class _BfDenseFn(torch.autograd.Function):
    """bf16 GEMM-input mode (BASELINE config 5): y = x W^T (+ b) with all three contractions on the library's bf16 MFMA kernels — forward
    tmjx_bgemm_nt against the weight's shadow, input gradient tmjx_bgemm_nt against the transposed shadow, weight + bias gradient
    tmjx_bgemm_dw (transposed LDS reads).  x and dy stay fp32 in memory and are converted when they are staged: no cast pass."""

    @staticmethod
    def forward(ctx, x, w, b, lin, dx_cols):
        x2 = _rows2d(x)
        if x2.data_ptr() % 16 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        sh = gemm_inputs.shadows
        ctx.save_for_backward(x2)
        ctx.lin, ctx.sh, ctx.params, ctx.dx_cols, ctx.x_shape = lin, sh, (w, b), dx_cols, x.shape
        N, K = w.shape
        return bgemm_nt(x2, sh.w[lin], N, K, b).view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        w, b = ctx.params
        N, K = w.shape
        dy2 = _rows2d(dy)
        if dy2.data_ptr() % 16 or dy2.stride(0) % 4:
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            cols = K if ctx.dx_cols is None else int(ctx.dx_cols)
            dx = torch.empty((dy2.shape[0], K), dtype=torch.float32, device=dy2.device)
            bgemm_nt(dy2, ctx.sh.wt[ctx.lin], cols, N, out=dx)            # (only the first `cols` columns are computed)
            dx = dx.view(ctx.x_shape)
        # straight into the flat gradient buffer's (row-padded) views where the learner provides them
        d = deferred_weight_grads.active
        gw = w.grad if (d is not None and w.grad is not None and w.grad.shape == w.shape and w.grad.stride(1) == 1 and id(w) not in d.seen) else None
        gb = b.grad if (b is not None and gw is not None and b.grad is not None) else None
        if gw is not None:
            d.seen.add(id(w))
        dw, db = bgemm_dw(dy2, x2, b is not None, out=gw, out_bias=gb)
        return dx, dw, db, None, None


class _Dense(nn.Linear):
    def forward(self, x):
        rows = x.numel() // x.shape[-1]
        if _bf16_ok(x, self):
            if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
                return _BfDenseFn.apply(x, self.weight, self.bias, self, None)
            return bgemm_nt(x, gemm_inputs.shadows.w[self], self.out_features, self.in_features, self.bias).view(*x.shape[:-1], self.out_features)
        if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32:      # fp32 mode, and the layers the bf16 kernels leave (1-wide head)
            if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
                return _HipDenseFn.apply(x, self.weight, self.bias, None)
            return gemm_nt(x, self.weight, self.bias).view(*x.shape[:-1], self.out_features)
        return F.linear(x, self.weight, self.bias)


def _dense(i: int, o: int, init=_lecun_uniform_) -> nn.Linear:
    lin = _Dense(i, o)
    init(lin.weight)
    nn.init.zeros_(lin.bias)
    return lin


class _SiluLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(silu(z + bias)) through tmjx_silu_ln_fwd / _bwd (csrc/ppo_kernels.h): one launch forward, two
    backward (dz plus d_gamma | d_beta | d_bias), instead of torch's bias-add, silu, layer_norm and five backward kernels."""

    @staticmethod
    def forward(ctx, z, bias, gamma, beta, eps):
        import ctypes as C
        from .. import hip as _hip
        H = z.shape[-1]
        rows = z.numel() // H
        z = z.contiguous()
        y = torch.empty_like(z)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=z.device)
        L = _hip.lib()
        with torch.cuda.device(z.device):
            stream = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
            _hip.check(L.tmjx_silu_ln_fwd(*[C.c_void_p(t.data_ptr()) for t in (z, bias, gamma, beta, y, stats)], rows, H, float(eps), stream),
                       "tmjx_silu_ln_fwd")
        ctx.save_for_backward(z, bias, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from .. import hip as _hip
        z, bias, gamma, stats = ctx.saved_tensors
        H = z.shape[-1]
        rows = z.numel() // H
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        grads = torch.empty((3, H), dtype=torch.float32, device=z.device)
        L = _hip.lib()
        partial = torch.empty(L.tmjx_silu_ln_partial_floats(rows, H), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            stream = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
            _hip.check(L.tmjx_silu_ln_bwd(*[C.c_void_p(t.data_ptr()) for t in (dy, z, bias, gamma, stats, dz, grads, partial)], rows, H, stream),
                       "tmjx_silu_ln_bwd")
        return dz, grads[2], grads[0], grads[1], None


def _block_fusable(x2, w) -> bool:
    """One-launch forward of a Dense -> SiLU -> LayerNorm block: layers exactly one GEMM tile wide (64 / 128 / 256) with aligned operands.
    Measured on MI355X (20480 rows): a first version that kept the accumulators' native layout (64-byte store pieces, 20 row statistics per lane)
    was SLOWER than GEMM + tmjx_silu_ln_fwd (47.1 vs 44.9 us at 256 x 256); with the transposed tile (four consecutive columns per lane: dwordx4
    stores of z and y, 5 row statistics per lane) the SGD half takes 69.9 ms against 70.4 ms.  TMJX_NO_FUSED_BLOCK=1 selects the two-launch form."""
    if not _hip_gemm_ok(x2, w) or os.environ.get("TMJX_NO_FUSED_BLOCK"):
        return False
    from .. import hip as _hip
    import ctypes as C
    return bool(_hip.lib().tmjx_gemm_nt_silu_ln_ok(C.c_void_p(x2.data_ptr()), x2.stride(0), C.c_void_p(w.data_ptr()), w.stride(0), w.shape[0]))


class _HipBlockFn(torch.autograd.Function):
    """y = LayerNorm(silu(x W^T + b)): forward ONE launch (tmjx_gemm_nt_silu_ln: the SiLU + LayerNorm epilogue on the accumulators of the
    MFMA tile, which spans whole rows); backward tmjx_silu_ln_bwd, then the input gradient (tmjx_gemm_nn) and the weight gradient
    (deferred into the grouped launch, or tmjx_gemm_dw).  Same saved tensors and arithmetic as _HipDenseFn + _SiluLayerNormFn."""

    @staticmethod
    def forward(ctx, x2, w, b, gamma, beta, eps, dx_cols):
        import ctypes as C
        from .. import hip as _hip
        M, K = x2.shape
        N = w.shape[0]
        z = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        y = torch.empty_like(z)
        stats = torch.empty((M, 2), dtype=torch.float32, device=x2.device)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(x2.device):
            _hip.check(_hip.lib().tmjx_gemm_nt_silu_ln(p(x2), x2.stride(0), p(w), w.stride(0), p(b), p(gamma), p(beta), p(z), p(y), N, p(stats), M, N, K,
                                                       float(eps), C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)), "tmjx_gemm_nt_silu_ln")
        ctx.save_for_backward(x2, w, z, b, gamma, stats)
        ctx.dx_cols, ctx.param = dx_cols, w
        ctx.producer = ln_bwd_links.producer_of(x2)
        ctx.link = None
        if ln_bwd_links.active:
            ctx.link = link = _BlockLink()
            link.y_ptr, link.y_shape, link.ctx = y.data_ptr(), tuple(y.shape), ctx
            ln_bwd_links.last = link
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        from .. import hip as _hip
        x2, w, z, b, gamma, stats = ctx.saved_tensors
        M, N = z.shape
        dy = dy.contiguous()
        if ctx.link is not None:
            ctx.link.ctx = None                     # (break the ctx <-> link cycle)
        if ctx.link is not None and ctx.link.dz_given:
            dz, grads = dy, ctx.link.grads          # the consumer's input-gradient GEMM already applied this block's LayerNorm + SiLU backward
            # consumed: a second backward through a retained graph finds link.ctx None, so its consumer sends a plain d loss / d y, which
            # must then take the block's own backward below
            ctx.link.dz_given, ctx.link.grads = False, None
        else:
            dz = torch.empty_like(z)
            grads = torch.empty((3, N), dtype=torch.float32, device=z.device)
            L = _hip.lib()
            partial = torch.empty(L.tmjx_silu_ln_partial_floats(M, N), dtype=torch.float32, device=z.device)
            with torch.cuda.device(z.device):
                _hip.check(L.tmjx_silu_ln_bwd(*[C.c_void_p(t.data_ptr()) for t in (dy, z, b, gamma, stats, dz, grads, partial)], M, N,
                                              C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)), "tmjx_silu_ln_bwd")
        if ctx.needs_input_grad[0] and _fusable_dx(dz, w, ctx.producer, ctx.dx_cols):
            dx = _dx_through_block(dz, w, ctx.producer)
        else:
            dx = gemm_nn(dz, w, ctx.dx_cols) if ctx.needs_input_grad[0] else None
        d = deferred_weight_grads.active
        got = d.try_add(dz, x2, ctx.param, None) if d is not None else None
        dw = got[0] if got is not None else gemm_dw(dz, x2, False)[0]
        return dx, dw, grads[2], grads[0], grads[1], None, None


class _Block(nn.Module):
    """Dense -> SiLU -> LayerNorm (flax LayerNorm defaults: eps 1e-6, scale + bias)."""
    FUSED_WIDTHS = (64, 128, 256, 512, 1024)

    def __init__(self, i: int, o: int):
        super().__init__()
        self.dense = _dense(i, o)
        self.norm = nn.LayerNorm(o, eps=1e-6)
        self.dx_cols = None       # set when only a column prefix of the input needs a gradient (the decoder's first block: the latent)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and self.dense.out_features in self.FUSED_WIDTHS and not torch.is_autocast_enabled():
            x2 = _rows2d(x)
            if torch.is_grad_enabled() and _block_fusable(x2, self.dense.weight):
                y = _HipBlockFn.apply(x2, self.dense.weight, self.dense.bias, self.norm.weight, self.norm.bias, self.norm.eps, self.dx_cols)
                return y.view(*x.shape[:-1], self.dense.out_features)
            if _bf16_ok(x2, self.dense):
                if torch.is_grad_enabled():
                    z = _BfDenseFn.apply(x2, self.dense.weight, None, self.dense, self.dx_cols)     # (the bias lives in the fused epilogue below)
                else:
                    z = bgemm_nt(x2, gemm_inputs.shadows.w[self.dense], self.dense.out_features, self.dense.in_features)
            elif not torch.is_grad_enabled():
                z = gemm_nt(x2, self.dense.weight)
            else:
                z = _HipDenseFn.apply(x2, self.dense.weight, None, self.dx_cols)     # (the bias lives in the fused epilogue below)
            y = _SiluLayerNormFn.apply(z, self.dense.bias, self.norm.weight, self.norm.bias, self.norm.eps)
            return y.view(*x.shape[:-1], self.dense.out_features)
        return self.norm(F.silu(self.dense(x)))


class _LatentConcatFn(torch.autograd.Function):
    """x = [mean + eps * exp(logvar / 2) | obs[:, ref:]] from fc2 = [mean | logvar] (reparameterize + the decoder-input concat,
    intention_network.py:78-88,128-139) as one launch forward (tmjx_latent_concat) and one backward (tmjx_latent_concat_bwd) instead
    of chunk / exp / mul / add / cat and their five backward kernels.  obs is not differentiated (network input)."""

    @staticmethod
    def forward(ctx, fc2, eps, obs, ref, handle=None):
        import ctypes as C
        from .. import hip as _hip
        ctx.handle = handle
        n, Z = eps.shape
        W = obs.shape[1]
        fc2, eps, obs = fc2.contiguous(), eps.contiguous(), obs.contiguous()
        wd = Z + W - ref
        # rows padded to a multiple of 4 floats (286 -> 288): the decoder's first GEMM reads this buffer with 16-byte row loads
        x = torch.empty((n, (wd + 3) // 4 * 4), dtype=torch.float32, device=fc2.device)     # (the pad columns are never read: the GEMM masks k >= K)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(fc2.device):
            _hip.check(_hip.lib().tmjx_latent_concat(p(fc2), p(eps), p(obs), p(x), n, Z, W, ref, obs.stride(0), obs.stride(1), None, None, x.shape[1], 0, None,
                                                     C.c_void_p(torch.cuda.current_stream(fc2.device).cuda_stream)), "tmjx_latent_concat")
        ctx.save_for_backward(fc2, eps)
        return x[:, :wd]

    @staticmethod
    def backward(ctx, dx):
        import ctypes as C
        from .. import hip as _hip
        fc2, eps = ctx.saved_tensors
        n, Z = eps.shape
        dx = dx.contiguous()
        dfc2 = torch.empty_like(fc2)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        # a second gradient of this fc2 that the caller wants summed into the one computed here instead of seeding autograd with it
        # (losses.ppo_loss_and_output_grads: the KL term's, from the loss head) — autograd would add the two with a launch of its own.  It
        # travels on THIS forward pass's handle (LatentGradHandle), not in process-wide state
        add = ctx.handle.take() if ctx.handle is not None else None
        if add is not None and (add.shape != fc2.shape or not add.is_contiguous()):
            raise RuntimeError("_LatentConcatFn: the pending fc2 gradient does not fit this forward pass")
        with torch.cuda.device(fc2.device):
            stream = C.c_void_p(torch.cuda.current_stream(fc2.device).cuda_stream)
            if add is None:
                _hip.check(_hip.lib().tmjx_latent_concat_bwd(p(dx), p(eps), p(fc2), p(dfc2), n, Z, dx.shape[1], stream), "tmjx_latent_concat_bwd")
            else:
                _hip.check(_hip.lib().tmjx_latent_concat_bwd_add(p(dx), p(eps), p(fc2), p(add), p(dfc2), n, Z, dx.shape[1], stream), "tmjx_latent_concat_bwd_add")
        return dfc2, None, None, None, None


class _LatentViewFn(torch.autograd.Function):
    """The decoder input x = [mean + eps * exp(logvar / 2) | proprioception] when the encoder chain's launch has ALREADY written it (tmjx_chain_fwd's latent
    tail): forward hands x on (no launch), backward is _LatentConcatFn's — d fc2 from d x through tmjx_latent_concat_bwd(_add)."""

    @staticmethod
    def forward(ctx, fc2, eps, x, handle=None):
        ctx.handle = handle
        ctx.save_for_backward(fc2.contiguous(), eps.contiguous())
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dx):
        return (_LatentConcatFn.backward(ctx, dx)[0], None, None, None)


class LatentGradHandle:
    """One policy forward pass's slot for a second gradient of its fc2 output (the KL term's): `policy.latent_grad_handle` after a forward that
    went through the fused latent kernel, else None.  `add(g)` parks the gradient, the backward of THAT forward's _LatentConcatFn node takes
    it.  A parked gradient that no backward pass consumed is an error at the policy's next forward (it would have been silently dropped)."""

    def __init__(self, fc2: torch.Tensor):
        self.ptr, self.shape, self._g = fc2.data_ptr(), tuple(fc2.shape), None

    def matches(self, fc2: torch.Tensor) -> bool:
        return fc2.data_ptr() == self.ptr

    def add(self, g: torch.Tensor) -> None:
        if self._g is not None:
            raise RuntimeError("LatentGradHandle: a gradient is already parked on this forward pass")
        self._g = g

    def take(self):
        g, self._g = self._g, None
        return g

    @property
    def pending(self) -> bool:
        return self._g is not None


class IntentionPolicy(nn.Module):
    """Encoder-decoder "intention" policy (intention_network.py:90-142)."""

    def __init__(self, obs_size: int, reference_obs_size: int, action_size: int, latents: int = 60,
                 encoder_layers: Sequence[int] = (1024, 1024), decoder_layers: Sequence[int] = (1024, 1024)):
        super().__init__()
        self.reference_obs_size, self.latents, self.action_size = reference_obs_size, latents, action_size
        enc, d = [], reference_obs_size
        for h in encoder_layers:
            enc.append(_Block(d, h)); d = h
        self.encoder = nn.Sequential(*enc)
        # fc2_mean and fc2_logvar of the reference (two Dense(latents), lecun_normal) as the two halves of ONE GEMM
        self.fc2 = _dense(d, 2 * latents, _lecun_normal_)
        dec, d = [], latents + (obs_size - reference_obs_size)
        for h in decoder_layers:
            dec.append(_Block(d, h)); d = h
        self.decoder = nn.Sequential(*dec)
        dec[0].dx_cols = latents      # the decoder input is [latent sample | proprioception]: only the latent part is differentiated
        self.head = _dense(d, 2 * action_size)

    def forward(self, obs: torch.Tensor, eps: torch.Tensor | None = None, deterministic: bool = False, return_fc2: bool = False):
        """obs already normalised. Returns (logits [.., 2*nu], latent_mean, latent_logvar), or (logits, mean | logvar as
        one [.., 2*latents] tensor) with `return_fc2` (what the fused loss head consumes)."""
        traj = obs[..., :self.reference_obs_size]
        prev, self.latent_grad_handle = getattr(self, "latent_grad_handle", None), None
        if prev is not None and prev.pending:
            raise RuntimeError("IntentionPolicy: the fc2 gradient parked on the previous forward pass (LatentGradHandle.add) was never consumed by a "
                               "backward pass through that forward — it would have been dropped")
        chains = _bf16_chain_ok(obs) and all(m.out_features % 4 == 0 for m in (self.fc2, self.head))
        if chains:                 # bf16 GEMM-input mode: the encoder (+ fc2) as ONE autograd function (fused epilogues, bf16 hidden activations)
            if getattr(self, "_enc_chain", None) is None:
                self._enc_chain = [_Layer("ln", b.dense, b.norm) for b in self.encoder] + [_Layer("dense", self.fc2)]
                self._dec_chain = [_Layer("ln", b.dense, b.norm) for b in self.decoder] + [_Layer("dense", self.head)]
            fc2 = bf16_chain(traj, self._enc_chain)
        else:
            if getattr(self, "_enc_f32", None) is None and obs.is_cuda:
                self._enc_f32 = [_Layer("ln", b.dense, b.norm) for b in self.encoder] + [_Layer("dense", self.fc2)]
                self._dec_f32 = [_Layer("ln", b.dense, b.norm) for b in self.decoder] + [_Layer("dense", self.head)]
            xlat = None
            if obs.is_cuda and _f32_chain_ok(traj, self._enc_f32):
                # 2 x 256 nets: the encoder + fc2 as ONE launch forward, one backward (csrc/mlp_chain.h).  The same launch CAN also sample the latent and write
                # the decoder's input (its latent tail, TMJX_CHAIN_LATENT=1; bit-identical) — measured 13 us per minibatch step SLOWER than the separate
                # tmjx_latent_concat launch (0.887 against 0.874 ms, two alternating pairs): that 25 us kernel runs next to the critic's chain on the other
                # stream, whereas the tail lengthens a kernel that holds every CU.  Off by default
                obs2 = obs.reshape(-1, obs.shape[-1])
                if (not deterministic and obs.dtype == torch.float32 and obs2.shape[0] * 2 * self.latents >= 2 * self.latents * 1024 and self.latents % 4 == 0
                        and (obs.shape[-1] - self.reference_obs_size) % 2 == 0 and obs2.stride(1) == 1 and obs2.stride(0) % 2 == 0 and self.reference_obs_size % 2 == 0
                        and os.environ.get("TMJX_CHAIN_LATENT") == "1"):
                    if eps is None:
                        eps = torch.randn(obs.shape[:-1] + (self.latents,), dtype=torch.float32, device=obs.device)
                    eps2 = eps.reshape(-1, self.latents).contiguous()
                    fc2, xlat = f32_chain(traj, self._enc_f32, latent=(eps2, obs2[:, self.reference_obs_size:]))
                else:
                    fc2 = f32_chain(traj, self._enc_f32)
            else:
                with ln_bwd_links():       # encoder and decoder are chains: each block's output feeds exactly one dense layer
                    h = self.encoder(traj)
                    fc2 = self.fc2(h)
        mean, logvar = torch.chunk(fc2, 2, dim=-1)
        if (not deterministic and fc2.is_cuda and fc2.dtype == torch.float32 and obs.dtype == torch.float32 and not torch.is_autocast_enabled()
                and fc2.numel() >= 2 * self.latents * 1024):
            # training-sized batches: latent sample + concat (and their backward) as one launch each
            lead = fc2.shape[:-1]
            if eps is None:
                eps = torch.randn_like(mean)
            handle = LatentGradHandle(fc2.reshape(-1, fc2.shape[-1])) if (fc2.requires_grad and fc2.is_contiguous()) else None
            if not chains and xlat is not None:
                x = _LatentViewFn.apply(fc2.reshape(-1, fc2.shape[-1]), eps.reshape(-1, self.latents), xlat, handle)
            else:
                x = _LatentConcatFn.apply(fc2.reshape(-1, fc2.shape[-1]), eps.reshape(-1, self.latents), obs.reshape(-1, obs.shape[-1]), self.reference_obs_size, handle)
            self.latent_grad_handle = handle
            x = x.view(*lead, x.shape[-1])
            if chains:
                logits = bf16_chain(x, self._dec_chain, dx_cols=self.latents)
            elif _f32_chain_ok(x, self._dec_f32, self.latents):
                logits = f32_chain(x, self._dec_f32, dx_cols=self.latents)
            else:
                with ln_bwd_links():
                    logits = self.head(self.decoder(x))
            if return_fc2:
                return logits, fc2
            return logits, mean, logvar
        if deterministic:
            z = mean
        else:
            if eps is None:
                eps = torch.randn_like(mean)
            z = mean + eps * torch.exp(0.5 * logvar)
        x = torch.cat([z, obs[..., self.reference_obs_size:]], dim=-1)
        logits = bf16_chain(x, self._dec_chain) if chains else self.head(self.decoder(x))
        if return_fc2:
            return logits, fc2
        return logits, mean, logvar


class _NoCtx:
    """Stand-in for an autograd context when a Function's forward is run outside autograd (inference)."""
    def save_for_backward(self, *a):
        pass


class ValueNet(nn.Module):
    """brax make_value_network: MLP(hidden..., 1), swish activations, lecun_uniform, output squeezed."""

    def __init__(self, obs_size: int, hidden: Sequence[int] = (1024, 1024)):
        super().__init__()
        layers, d = [], obs_size
        for h in hidden:
            layers += [_dense(d, h), nn.SiLU()]; d = h
        layers.append(_dense(d, 1))
        self.net = nn.Sequential(*layers)

    def forward(self, obs):
        if _bf16_chain_ok(obs) and len(self.net) > 1:
            # bf16 GEMM-input mode: the hidden layers as one chain (Dense -> SiLU epilogues), the 1-wide head on the fp32 kernels
            if getattr(self, "_chain", None) is None:
                dense = [m for m in self.net if isinstance(m, nn.Linear)]
                self._chain = [_Layer("silu", m) for m in dense[:-1]]
                head = dense[-1]
                if head.out_features == 1 and head.in_features % 4 == 0 and head.in_features <= 1024 and not os.environ.get("TMJX_NO_CHAIN_HEAD"):
                    self._chain.append(_Layer("head", head))         # the 1-wide head inside the chain: its input gradient is never materialised
            h = bf16_chain(obs, self._chain, last_y_f32=True)
            if self._chain[-1].kind == "head":
                return h.squeeze(-1)
            return self.net[-1](h).squeeze(-1)
        if obs.is_cuda and obs.dtype == torch.float32 and gemm_inputs.dtype is None:
            # fp32 on the GPU: every hidden layer is ONE launch forward (GEMM + SiLU epilogue), no torch element-wise kernel in either direction
            h = obs
            dense = [m for m in self.net if isinstance(m, nn.Linear)]
            if len(dense) > 1 and dense[-1].out_features == 1:
                if getattr(self, "_f32_chain", None) is None:
                    self._f32_chain = [_Layer("silu", m) for m in dense[:-1]] + [_Layer("head", dense[-1])]
                if os.environ.get("TMJX_VALUE_CHAIN", "1") != "0" and _f32_chain_ok(obs, self._f32_chain):
                    # 256-wide critic: the whole MLP incl. its 1-wide head as ONE launch forward, one backward (csrc/mlp_chain.h) — in the learner's pass
                    # and in the bootstrap value's alike (bit-identical to the layer-by-layer functions below)
                    return f32_chain(obs, self._f32_chain).squeeze(-1)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and _value_chain_ok(_rows2d(obs), dense):
                # the learner's pass: the whole MLP as one autograd function (no element-wise launch between the backward GEMMs, the 1-wide head without a GEMM)
                return _ValueChainFn.apply(obs, dense, *[p for lin in dense for p in (lin.weight, lin.bias)]).squeeze(-1)
            for lin in dense[:-1]:
                if torch.is_grad_enabled() and (lin.weight.requires_grad or h.requires_grad):
                    h = _HipSiluDenseFn.apply(h, lin.weight, lin.bias)
                else:
                    h = _HipSiluDenseFn.forward(_NoCtx(), h, lin.weight, lin.bias)
            return dense[-1](h).squeeze(-1)
        return self.net(obs).squeeze(-1)


class NormalTanh:
    """NormalTanhDistribution over `logits = [loc, raw_scale]` (min_std 0.001)."""
    MIN_STD = 0.001
    LOG2 = math.log(2.0)

    @staticmethod
    def params(logits):
        loc, raw = torch.chunk(logits, 2, dim=-1)
        return loc, F.softplus(raw) + NormalTanh.MIN_STD

    @staticmethod
    def sample_no_postprocessing(logits, noise=None):
        loc, scale = NormalTanh.params(logits)
        if noise is None:
            noise = torch.randn_like(loc)
        return loc + scale * noise

    @staticmethod
    def _fldj(x):  # log |d tanh(x)/dx|
        return 2.0 * (NormalTanh.LOG2 - x - F.softplus(-2.0 * x))

    @staticmethod
    def log_prob(logits, raw_action):
        loc, scale = NormalTanh.params(logits)
        lp = -0.5 * ((raw_action - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)
        return (lp - NormalTanh._fldj(raw_action)).sum(-1)

    @staticmethod
    def entropy(logits, noise=None):
        loc, scale = NormalTanh.params(logits)
        ent = 0.5 + 0.5 * math.log(2 * math.pi) + torch.log(scale)
        x = NormalTanh.sample_no_postprocessing(logits, noise)
        return (ent + NormalTanh._fldj(x)).sum(-1)

    @staticmethod
    def postprocess(raw_action):
        return torch.tanh(raw_action)

    @staticmethod
    def mode(logits):
        return torch.tanh(torch.chunk(logits, 2, dim=-1)[0])


class RunningStatistics:
    """Batched Welford observation normaliser (reference math: agent/masked_running_statistics.py:95-236;
    the learner calls brax's copy of the same code at mlp_ppo/ppo.py:357-361,503-505).
    Across ranks the per-column sums are all-reduced once per update (same totals as the reference's three psums)."""

    def __init__(self, size: int, device, std_min: float = 1e-6, std_max: float = 1e6):
        self.count = torch.zeros((), dtype=torch.float32, device=device)
        self.mean = torch.zeros(size, dtype=torch.float32, device=device)
        self.summed_variance = torch.zeros(size, dtype=torch.float32, device=device)
        self.std = torch.ones(size, dtype=torch.float32, device=device)
        self.std_min, self.std_max = std_min, std_max

    @torch.no_grad()
    def update(self, batch: torch.Tensor, group=None, distributed: bool | None = None) -> None:
        """One pass over the batch: per column S1 = sum(x - mean_old), S2 = sum((x - mean_old)^2); then
        mean_update = S1 / count_new and variance_update = sum((x - mean_old)(x - mean_new)) = S2 - mean_update * S1 — the
        reference's update (masked_running_statistics.py:161-214) with its second pass over the data folded into the first.
        `distributed` (default: a process group is initialised with more than one rank): S1 | S2 are summed across the ranks of
        `group` (None = the default group) in ONE all-reduce — the same totals as the reference's psums of the count, mean_update
        and variance_update (ppo.py:357-361).  On the GPU the sums and the in-place update are HIP kernels (tmjx_stats_sums /
        tmjx_stats_apply, K6); CPU tensors (tests, gloo) take the same formulas in torch."""
        import torch.distributed as dist
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        flat = batch.reshape(-1, batch.shape[-1])
        if not flat.is_contiguous():
            flat = flat.contiguous()
        rows, W = flat.shape
        n_added = float(rows * (dist.get_world_size(group) if distributed else 1))      # equal shards (shard_range)
        if flat.is_cuda and flat.dtype == torch.float32 and W % 4 == 0:
            import ctypes as C
            from .. import hip as _hip
            L = _hip.lib()
            p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            if getattr(self, "_sums", None) is None:
                self._sums = torch.empty(2 * W, dtype=torch.float32, device=flat.device)
                self._scratch = torch.empty(L.tmjx_stats_scratch_floats(W), dtype=torch.float32, device=flat.device)
            with torch.cuda.device(flat.device):
                stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
                _hip.check(L.tmjx_stats_sums(p(flat), p(self.mean), p(self._sums), p(self._scratch), rows, W, stream), "tmjx_stats_sums")
                if distributed:
                    dist.all_reduce(self._sums, group=group)
                _hip.check(L.tmjx_stats_apply(p(self._sums), n_added, p(self.count), p(self.mean), p(self.summed_variance), p(self.std), W,
                                              float(self.std_min), float(self.std_max), stream), "tmjx_stats_apply")
            return
        d = flat - self.mean
        sums = torch.cat([d.sum(0), (d * d).sum(0)])
        if distributed:
            dist.all_reduce(sums, group=group)
        count = self.count + n_added
        upd = sums[:W] / count
        # in place: the SGD-loop graph (PPOLearner) holds pointers to these buffers
        self.summed_variance.add_(sums[W:] - upd * sums[:W])
        self.mean.add_(upd); self.count.copy_(count)
        self.std.copy_(torch.sqrt(torch.clamp(self.summed_variance, min=0) / count).clamp(self.std_min, self.std_max))

    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean) / self.std

    def state_dict(self):
        return {"count": self.count, "mean": self.mean, "summed_variance": self.summed_variance, "std": self.std}
