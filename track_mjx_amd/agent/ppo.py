"""PPO training loop for the tracking task — mirror of track_mjx/agent/mlp_ppo/ppo.py:128-809.

Structure of the reference (ppo.py:279-441): training_epoch = scan(training_step); training_step =
  (batch_size*num_minibatches/num_envs) x generate_unroll(unroll_length)  -> normaliser update ->
  num_updates_per_batch x [one row permutation -> num_minibatches x (loss, grad, pmean, clip(10), adam)].
Here: one process per GPU; envs are sharded across ranks (each rank owns num_envs/world envs and its share of
every minibatch, exactly the reference's pmap split ppo.py:477-480,306-311); gradients are all-reduced (mean) with
RCCL once per minibatch step (reference collective C1, gradients.gradient_update_fn(pmap_axis_name)), the
normaliser statistics with three small sum all-reduces per training step (C2).
"""
from __future__ import annotations

import logging
import math
import time
from typing import Callable

import numpy as np
import torch
import torch.distributed as dist

from . import losses as _losses
from .networks import Bf16Shadows, IntentionPolicy, NormalTanh, RunningStatistics, ValueNet, deferred_weight_grads, gemm_inputs


import contextlib
import os
_nullctx = contextlib.nullcontext


def _flat_layout(params):
    """Segments of the flat parameter / gradient / moment buffers: every tensor starts on a 16-byte boundary and a weight matrix whose
    row length is not a multiple of 4 (the 470- and 286-wide first layers) gets its rows padded to one: the MFMA GEMM kernels then
    take the parameter itself through their vector-load path (leading dimension 472 / 288, the true K masks the pad), and the pad
    columns stay exactly zero (zero gradient, zero moments).  Returns [(offset, rows, cols, padded_cols)], total floats."""
    segs, off = [], 0
    for p in params:
        if p.dim() == 2 and p.shape[1] % 4 and not os.environ.get("TMJX_NO_PAD"):
            rows, cols, pc = p.shape[0], p.shape[1], (p.shape[1] + 3) // 4 * 4
        else:
            rows, cols, pc = 1, p.numel(), p.numel()
        segs.append((off, rows, cols, pc))
        off += (rows * pc + 3) // 4 * 4
    return segs, off


def _flat_view(flat, seg, like):
    off, rows, cols, pc = seg
    if pc == cols:
        return flat[off:off + rows * cols].view_as(like)
    return flat[off:off + rows * pc].view(rows, pc)[:, :cols]


class FlatGrads:
    """All parameter gradients live in ONE contiguous fp32 buffer, so the data-parallel mean is a single
    all-reduce of 2.5-17 MB (sized for the 7 x 153 GB/s xGMI links: one large message instead of per-tensor
    calls) and global-norm clipping is one reduction."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.segs, n = _flat_layout(self.params)
        self.flat = torch.zeros(n, dtype=self.params[0].dtype, device=self.params[0].device)
        self._avg_ok = None
        for p, seg in zip(self.params, self.segs):
            p.grad = _flat_view(self.flat, seg, p)

    def zero(self):
        self.flat.zero_()

    def assign(self, grads):
        """Write freshly computed gradients (torch.autograd.grad) into the flat buffer with one multi-tensor copy: no
        zero-fill and no per-parameter `grad += new` kernels as with loss.backward() into pre-existing .grad views."""
        grads = list(grads)
        todo = [i for i, p in enumerate(self.params) if grads[i].data_ptr() != p.grad.data_ptr()]     # (deferred weight gradients are already in place)
        dense = [i for i in todo if self.params[i].grad.is_contiguous()]
        # one multi-tensor kernel for the dense views; the row-padded (strided) ones separately — a single strided destination sends the
        # WHOLE foreach call down its per-tensor runtime-copy path (25 copyBuffer nodes per SGD step, +0.12 ms)
        if dense:
            torch._foreach_copy_([self.params[i].grad for i in dense], [grads[i] for i in dense])
        for i in todo:
            if not self.params[i].grad.is_contiguous():
                self.params[i].grad.copy_(grads[i])

    def assign_subset(self, params, grads):
        """`assign` for a subset of the parameters (one network's backward pass: PPOLearner's bucketed gradient all-reduce)."""
        todo = [(p, g) for p, g in zip(params, grads) if g.data_ptr() != p.grad.data_ptr()]
        dense = [(p, g) for p, g in todo if p.grad.is_contiguous()]
        if dense:
            torch._foreach_copy_([p.grad for p, _ in dense], [g for _, g in dense])
        for p, g in todo:
            if not p.grad.is_contiguous():
                p.grad.copy_(g)

    def all_reduce_mean(self, group=None, force: bool = False, lo: int | None = None, hi: int | None = None, async_op: bool = False):
        """`force`: issue the collective on a one-rank group too (identity) — lets a single-GPU box exercise the RCCL call path.
        `lo:hi`: one BUCKET of the flat buffer (the policy's or the value network's gradients); `async_op`: return the work handle(s) instead of
        waiting — the caller overlaps the collective with the other network's backward pass and calls `.wait()` in front of the optimiser."""
        works = []
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
            buf = self.flat if lo is None else self.flat[lo:hi]
            if self._avg_ok is None:       # RCCL reduces with ncclAvg itself (no separate division launch); gloo (CPU tests) has no AVG
                try:
                    probe = torch.ones(1, dtype=self.flat.dtype, device=self.flat.device)
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                    self._avg_ok = bool(probe.item() == 1.0)
                except Exception:  # noqa: BLE001 — backend without AVG: sum, then divide
                    self._avg_ok = False
            if self._avg_ok:
                w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
            else:
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
                if async_op:
                    w.wait()               # (gloo: CPU tests — the division needs the sum)
                    w = None
                buf.div_(dist.get_world_size(group))
            if async_op and w is not None:
                works.append(w)
        return works

    def clip_by_global_norm(self, max_norm: float):
        norm = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / torch.clamp(norm, min=max_norm), max=1.0))  # scale = max_norm / max(max_norm, norm)
        return norm


class FlatAdam:
    """optax.chain(clip_by_global_norm(max_norm), adam(lr)) (reference: ppo.py:517-520) on the flat buffers: the parameters are
    re-seated as views of ONE contiguous fp32 buffer (same layout as the gradients of FlatGrads), so the optimiser step is a norm reduction
    plus one launch of tmjx_adam_clip instead of the clip kernels, a multi-tensor scale and a multi-tensor Adam over ~30 tensors.
    Must be built before any hipGraph captures the parameters' addresses.  CPU tensors (tests) take the same maths in torch."""

    def __init__(self, grads: FlatGrads, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 10.0):
        self.grads, self.lr, self.betas, self.eps, self.max_norm = grads, lr, betas, eps, max_norm
        flat = torch.zeros_like(grads.flat)
        with torch.no_grad():
            for p, seg in zip(grads.params, grads.segs):
                view = _flat_view(flat, seg, p)
                view.copy_(p.detach())
                p.data = view
        self.flat = flat
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
        self.t = 0

    @torch.no_grad()
    def step(self):
        g = self.grads.flat
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        if g.is_cuda:
            # the global norm is part of the step: per-workgroup sums of squares (one launch, fixed order: bit-identical on every rank) that the
            # Adam kernel adds up itself — no library reduction kernel (tmjx_adam_clip_norm)
            import ctypes as C
            from .. import hip as _hip
            if getattr(self, "_norm_scratch", None) is None:
                self._norm_scratch = torch.empty(int(_hip.lib().tmjx_adam_norm_floats()), dtype=torch.float32, device=g.device)
                self._norm = torch.zeros((), dtype=torch.float32, device=g.device)
            norm = self._norm
            with torch.cuda.device(g.device):
                _hip.check(_hip.lib().tmjx_adam_clip_norm(*[C.c_void_p(t.data_ptr()) for t in (self.flat, g, self.exp_avg, self.exp_avg_sq, self._norm_scratch, norm)],
                                                          g.numel(), self.lr, b1, b2, self.eps, bc1, bc2, self.max_norm,
                                                          C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)), "tmjx_adam_clip_norm")
        else:
            norm = torch.linalg.vector_norm(g)
            gs = g * (self.max_norm / torch.clamp(norm, min=self.max_norm))
            self.exp_avg.mul_(b1).add_(gs, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(gs, gs, value=1 - b2)
            self.flat.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / (bc2 ** 0.5) + self.eps, value=-self.lr / bc1)
        return norm

    def state_dict(self):
        return {"t": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}


RESIDENT_ENVS_PER_CU = 12      # three waves per SIMD (csrc/tmjx_wave.hip: the register bound; the LDS image, 9 granules of 1 280 B since round 5, would allow 14)


def default_groups(n_envs: int, device=None, widest_layer: int = 256) -> int:
    """How many env groups collect() should pipeline: THREE.  While one group is in its serial reward / observation / inference phase the
    others should still fill the GPU's resident-env slots (12 per CU: 3072 on an MI355X; 2816 when this was measured): two of three groups
    of 4096 envs fill 89 % (97 % of 2816), one of two 67 % (73 %); re-measured in round 4 at 12 per CU: equal thirds 157.9 ms, 1536 / 1536 / 1024
    159.4 ms, two groups 169.1 ms, four 263.7 ms (tools/group_sizes_ab.sh) (config 2: roll-out 170.1 ms with two groups, 159.8 ms with three; config 5, 8192 envs: 345.8 -> 330.5 ms; config 4's wide
    policy: 194.2 / 195.3 ms, a tie).  Never more than three: see group_sizes.  (`device`, `widest_layer`: kept for callers that tune per
    model; both configurations that once preferred two groups — 8192 envs, the wide policy — stopped doing so when the acting path's
    element-wise kernels went to one-wave blocks.)"""
    return 1 if n_envs < 8 else 3


def group_sizes(n_envs: int, n_groups: int) -> list[int]:
    """A rank's envs as `n_groups` env groups for the pipelined roll-out (collect()): as equal as multiples of 4 allow, the remainder in the
    first groups (4096 envs, 3 groups -> 1368, 1364, 1364).  Three groups, not two: while one group is in its serial reward / observation /
    inference phase the other two still fill 97 % of the GPU's 2816 resident-env slots (two groups: 73 %) — roll-out 173.6 -> 166.3 ms at
    config 2.  Not four: with the default stream that is more HIP streams than hardware queues, and groups that share a queue serialise
    (287 ms)."""
    n_groups = max(1, min(int(n_groups), n_envs // 4 if n_envs >= 4 else 1))
    base = (n_envs // n_groups) // 4 * 4
    if base == 0:
        return [n_envs]
    sizes = [base] * n_groups
    rest = n_envs - base * n_groups
    k = 0
    while rest > 0:
        add_ = min(4, rest)
        sizes[k % n_groups] += add_
        rest -= add_
        k += 1
    return sizes


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous env range [lo, hi) owned by `rank` (reference: reshape (local_devices, num_envs/devices), ppo.py:477-480)."""
    if total % world:
        raise ValueError(f"num_envs={total} must be divisible by the number of ranks ({world})")
    per = total // world
    return rank * per, (rank + 1) * per


class PPOLearner:
    def __init__(self, env, *, encoder_layers, decoder_layers, critic_layers, latents: int = 60, learning_rate: float = 1e-4,
                 entropy_cost: float = 1e-2, discounting: float = 0.98, reward_scaling: float = 1.0, gae_lambda: float = 0.95,
                 clipping_epsilon: float = 0.2, unroll_length: int = 20, batch_size: int = 1024, num_minibatches: int = 16,
                 num_updates_per_batch: int = 4, normalize_observations: bool = True, kl_weight: float = 0.1,
                 seed: int = 0, group=None, matmul_dtype: torch.dtype | None = None, use_graph: bool = True, shuffle_rng: str = "torch",
                 act_rng: str = "device"):
        # `env` may be a LIST of envs (equal halves of this rank's envs): their roll-outs are then pipelined on one HIP stream
        # each (collect()), so that the tail of one half's physics kernel, its reward / observation kernels and its policy
        # inference run next to the other half's physics kernel
        self.envs = list(env) if isinstance(env, (list, tuple)) else [env]
        env = self.envs[0]
        self.env, self.group = env, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # C1 / C2 are issued when there is more than one rank; TMJX_COLLECTIVES_ALWAYS=1 also issues them on a one-rank process group
        # (identities), so that the RCCL + hipGraph interplay of the training step can be run on a single-GPU box (tests/test_gpu_rccl.py)
        self.collectives = self.world > 1 or bool(dist.is_available() and dist.is_initialized() and os.environ.get("TMJX_COLLECTIVES_ALWAYS"))
        dev = env.device
        self.dev = dev
        n_local = sum(e.num_envs for e in self.envs)
        self.n_local = n_local
        self.num_envs_global = n_local * self.world
        if (batch_size * num_minibatches) % self.num_envs_global:
            raise AssertionError(f"batch_size*num_minibatches % num_envs = {(batch_size * num_minibatches) % self.num_envs_global}")
        self.unrolls = batch_size * num_minibatches // self.num_envs_global
        self.T, self.num_minibatches, self.num_updates = unroll_length, num_minibatches, num_updates_per_batch
        self.local_batch = batch_size // self.world
        self.hp = dict(entropy_cost=entropy_cost, discounting=discounting, reward_scaling=reward_scaling, gae_lambda=gae_lambda,
                       clipping_epsilon=clipping_epsilon)
        self.kl_weight = kl_weight
        self.normalize_observations = normalize_observations
        self.env_steps_per_training_step = batch_size * unroll_length * num_minibatches
        obs, ref = env.observation_size, int(env.layout.ref_obs_size)
        torch.manual_seed(seed)  # identical init on every rank (reference: device_put_replicated, ppo.py:625-627)
        self.policy = IntentionPolicy(obs, ref, env.action_size, latents, encoder_layers, decoder_layers).to(dev)
        self.value = ValueNet(obs, critic_layers).to(dev)
        self.params = list(self.policy.parameters()) + list(self.value.parameters())
        self.grads = FlatGrads(self.params)
        self._n_policy_params = len(list(self.policy.parameters()))
        # C1 in TWO buckets (the flat buffer is [policy | value]): the value network's gradients are reduced while the policy's backward pass
        # still runs (update()).  The reference's pmean sits inside the jitted step where XLA overlaps it (ppo.py:621-623)
        self._bucket_split = self.grads.segs[self._n_policy_params][0]
        # (the value network's gradient views, and the workgroup budget of an EARLY weight-gradient group for them on the side stream, behind that
        # network's backward pass and next to the policy's: TMJX_VALUE_DW_WGS.  Default 0 = one group at the end of the step: measured in round 5 at
        # budgets 256 / 384 / 512 / 768 — 0.99 – 1.02 ms per config-2 minibatch step against 0.97 – 1.00 for the single group; the chip is busy either way)
        self._value_grad_ptrs = {p.grad.data_ptr() for p in self.grads.params[self._n_policy_params:]}
        self._value_dw_wgs = int(os.environ.get("TMJX_VALUE_DW_WGS", "0"))
        # Used when the gradient buffer is large (>= 8 MB: the rodent-mc-intention nets' 17.2 MB, ~ 0.2 ms on a ring) — cutting the captured
        # step into three graphs and issuing two collectives costs ~ 0.15 ms per minibatch step (measured with a one-rank RCCL group,
        # profiles/r04_bench_selflaunch_one_rank_bucketed.json), more than the 2.49 MB buffer of the 2x256 nets takes to reduce.  TMJX_BUCKET_OVERLAP=1 / 0 forces it
        big = self.grads.flat.numel() * self.grads.flat.element_size() >= 8 << 20
        force = os.environ.get("TMJX_BUCKET_OVERLAP")
        # (by default only with MORE than one rank: a one-rank group has nothing to overlap and pays the three-graph form's 0.15 ms)
        self.overlap_c1 = self.collectives and ((big and self.world > 1) if force is None else force == "1")
        self.opt = FlatAdam(self.grads, learning_rate, betas=(0.9, 0.999), eps=1e-8, max_norm=10.0)   # optax.clip_by_global_norm(10.0) -> adam
        self.normalizer = RunningStatistics(obs, dev)
        self.gen = torch.Generator(device=dev).manual_seed(seed * 1000 + 17 + self.rank)
        self.gens = [self.gen] + [torch.Generator(device=dev).manual_seed(seed * 1000 + 17 + self.rank + 7919 * g) for g in range(1, len(self.envs))]
        rows = self.unrolls * n_local
        T = self.T
        f32 = dict(dtype=torch.float32, device=dev)
        self.buf = {"observation": torch.empty((T, rows, obs), **f32), "raw_action": torch.empty((T, rows, env.action_size), **f32),
                    "log_prob": torch.empty((T, rows), **f32), "reward": torch.empty((T, rows), **f32),
                    "discount": torch.empty((T, rows), **f32), "truncation": torch.empty((T, rows), **f32),
                    "next_observation_last": torch.empty((rows, obs), **f32)}
        self.matmul_dtype = matmul_dtype
        # bf16 GEMM-input mode (BASELINE config 5): bf16 shadows of every dense layer's weight (built AFTER FlatAdam re-seated the parameters
        # in the flat buffer), refreshed at the start of every SGD step; the first layers need no transposed shadow (no input gradient)
        self.shadows = None
        if matmul_dtype == torch.bfloat16 and dev.type == "cuda":
            lins = [m for net in (self.policy, self.value) for m in net.modules() if isinstance(m, torch.nn.Linear) and m.out_features % 4 == 0]
            first = {self.policy.encoder[0].dense, next(m for m in self.value.net if isinstance(m, torch.nn.Linear))}
            self.shadows = Bf16Shadows(lins, need_t=[m for m in lins if m not in first])
        # ... and a bf16 twin of the minibatch's normalised observations, written by the gather launch (tmjx_minibatch_begin_bf16): [rows][ceil64(obs)],
        # zero beyond the observation width for good (the launch never writes there)
        self._obs16 = None
        if self.shadows is not None and obs % 4 == 0 and not os.environ.get("TMJX_NO_BF16_TWIN"):
            self._obs16 = torch.zeros((T * self.local_batch, (obs + 63) // 64 * 64), dtype=torch.bfloat16, device=dev)
        self._sgd_side = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and os.environ.get("TMJX_SGD_TWO_STREAMS", "1") != "0") else None
        if self._sgd_side is not None and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)    # intentional: the value net's gradients arrive from the side stream
        self._metric_index = torch.tensor([0, 1, 2, 4, 3], dtype=torch.long, device=dev)    # METRIC_KEYS -> slots of tmjx_ppo_loss's output
        self.use_graph, self._graph, self._graph_kl, self._graph_selfadv = use_graph, None, None, None
        self._split_graphs = None          # (forward + loss head, value backward, policy backward) of the bucketed step
        self._c1_side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # self-advancing SGD step (tmjx_minibatch_begin): the epoch's permutation, {draw counter, slot, ticket}, the metric accumulator
        self._perm_static = torch.zeros(rows, dtype=torch.long, device=dev)
        self._mb_state = torch.zeros(16 + 16 * 64, dtype=torch.long, device=dev)     # TMJX_MINIBATCH_STATE_WORDS: {draw counter, slot, tickets ...}
        self._acc8 = torch.zeros(8, dtype=torch.float32, device=dev)
        self._noise_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019 * (self.rank + 1)) & (2 ** 64 - 1)      # one Philox key per rank
        self._act_graphs: dict = {}
        self._wpad: dict = {}
        self.lds_free = len(self.envs) > 1 and dev.type == "cuda"   # pipelined roll-outs: LDS-free inference kernels (see _act_fused)
        # minibatch shuffle: "torch" = torch.randperm on the device (default, nothing leaves the GPU); "jax" = the reference's own draws from
        # the seed (jax_random.SgdKeys: key plumbing of ppo.py:443-451,303-307,324 + jax.random.permutation), computed on the host
        self.sgd_keys = None
        # acting noise: "device" (default) = Philox draws inside the inference kernels; "torch" = the learner's torch generators; "jax"
        # (needs shuffle_rng="jax") = the reference's own draws from its key plumbing
        self.act_rng = act_rng
        self._act_rng: dict = {}
        if act_rng not in ("device", "torch", "jax") or (act_rng == "jax" and shuffle_rng != "jax"):
            raise ValueError("act_rng must be 'device', 'torch' or 'jax' (the latter together with shuffle_rng='jax')")
        if shuffle_rng == "jax":
            from ..jax_random import SgdKeys
            self.sgd_keys = SgdKeys(seed, process_id=0, device_index=self.rank, local_devices=self.world)
            self.perm_fn = lambda upd, rows: torch.from_numpy(self.sgd_keys.permutation(rows).astype("int64"))
        elif shuffle_rng != "torch":
            raise ValueError("shuffle_rng must be 'torch' or 'jax'")
        # the roll-out generators' device-side Philox streams exist from the start, in group order (checkpoint.py saves / restores their
        # counters in place: a captured inference graph keeps reading the same tensors); other generators (the evaluator's) join lazily
        for gen in self.gens:
            self._act_rng_state(gen)
        self.states = [None] * len(self.envs)
        # LDS-free acting path: a ROW-major staging copy of every group's newest observation (third destination of the roll-out store): the acting
        # policy's first layer then reads 16 bytes along K per lane instead of four strided words from the env's [obs][n_env] buffer
        self._obs_rm = ([torch.zeros((e.num_envs, e.observation_size), dtype=torch.float32, device=dev) for e in self.envs]
                        if (self.lds_free and env.observation_size % 4 == 0 and not os.environ.get("TMJX_NO_OBS_STAGING")) else None)
        self._streams = [torch.cuda.Stream(device=dev) for _ in self.envs] if (len(self.envs) > 1 and dev.type == "cuda") else None

    @property
    def state(self):
        return self.states[0]

    @state.setter
    def state(self, st):
        self.states[0] = st

    def n_params(self) -> int:
        return int(sum(p.numel() for p in self.grads.params))

    # ---- acting (brax acting.generate_unroll / actor_step through make_inference_fn, ppo_networks.py:46-96)
    @torch.no_grad()
    def act(self, obs: torch.Tensor, deterministic: bool = False, gen: torch.Generator | None = None, draws=None):
        """`draws` = (eps [n, latents], noise [n, A]) replaces the generator's normal draws (same-seed mode: the reference's own noise)."""
        gen = self.gen if gen is None else gen
        if draws is not None:
            x = self.normalizer.normalize(obs) if self.normalize_observations else obs
            logits, mean, logvar = self.policy(x, eps=draws[0], deterministic=False)
            logits = logits.float()
            raw = NormalTanh.sample_no_postprocessing(logits, draws[1])
            return NormalTanh.postprocess(raw), {"raw_action": raw, "log_prob": NormalTanh.log_prob(logits, raw), "logits": logits,
                                                 "latent_mean": mean, "latent_logvar": logvar}
        # (the acting policy runs on the fp32 LDS-free kernels in either GEMM-input mode: next to the physics kernel the matrix pipe is idle)
        if (not deterministic and self.dev.type == "cuda" and obs.dim() == 2 and obs.dtype == torch.float32):
            # the RAW observation: the env's [obs][n_env] buffer (K-major float4 loads need 4 | n_env) or a row-major copy of it (collect()'s staging
            # buffer, written by the roll-out store: 16-byte loads along K — the first layer read K-major took 44 us of a group's 338 us serial
            # phase at config 2, and 530 us at config 5's 1024-wide first layer)
            if self.lds_free and ((obs.shape[0] % 4 == 0 and obs.stride(0) == 1) or (obs.stride(1) == 1 and obs.stride(0) % 4 == 0 and obs.data_ptr() % 16 == 0)):
                return self._act_fused(None, obs_raw=obs, gen=gen)
            return self._act_fused(self.normalizer.normalize(obs) if self.normalize_observations else obs, gen=gen)
        x = self.normalizer.normalize(obs) if self.normalize_observations else obs
        eps = torch.randn((x.shape[0], self.policy.latents), generator=gen, device=self.dev)
        logits, mean, logvar = self.policy(x, eps=eps, deterministic=deterministic)
        logits = logits.float()
        if deterministic:
            return NormalTanh.mode(logits), {"latent_mean": mean, "latent_logvar": logvar}
        noise = torch.randn((x.shape[0], self.policy.action_size), generator=gen, device=self.dev)
        raw = NormalTanh.sample_no_postprocessing(logits, noise)
        return NormalTanh.postprocess(raw), {"raw_action": raw, "log_prob": NormalTanh.log_prob(logits, raw), "logits": logits,
                                             "latent_mean": mean, "latent_logvar": logvar}

    def _act_fused(self, x: torch.Tensor, obs_raw: torch.Tensor | None = None, gen: torch.Generator | None = None):
        """Stochastic inference with the latent sample + decoder-input concat and the action sample / tanh / log-prob as one
        HIP kernel each (tmjx_latent_concat, tmjx_sample_action) instead of ~25 element-wise launches.  Same random draws, in
        the same order, as the torch path of act().

        With `self.lds_free` (pipelined roll-outs) every dense layer goes through tmjx_linear_nolds (`obs_raw` = the env's
        observation buffer, normalised here element-wise): no kernel of the inference touches LDS, so it runs on its stream NEXT TO the other half's physics kernel, which owns every CU's LDS."""
        import ctypes as C
        from .. import hip as _hip
        gen = self.gen if gen is None else gen
        pol, L = self.policy, _hip.lib()
        Z, A, ref = pol.latents, pol.action_size, pol.reference_obs_size
        lds_free = self.lds_free and obs_raw is not None
        # LDS-free path: the observation is read RAW from the env's [obs][n_env] buffer — the normalisation (obs - mean) / std is applied while the
        # first encoder layer loads its operand (tmjx_linear_nolds_norm; 1 / std refreshed once per collect(): _refresh_padded_weights) and by
        # tmjx_latent_concat for the proprioceptive part: no element-wise launch in front of the inference.  (Folding it into the layer's
        # weights instead — W / std, b - (W / std) mean — was tried and is NOT safe: a near-constant observation column has std = 1e-6, and
        # x W / std - mean W / std then cancels catastrophically: 0.24 absolute error on the logits in the unit test)
        fold = lds_free and self.normalize_observations
        if fold and getattr(self, "_fold", None) is None:
            self._refresh_padded_weights()
        if lds_free:
            x = obs_raw
        src = x
        n, W = src.shape
        f32 = dict(dtype=torch.float32, device=self.dev)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(self.dev):
            stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

            # bf16 GEMM-input mode (BASELINE config 5): the acting policy's layers take the SAME operand rounding as the learner's forward pass
            # (bf16 activations x the resident bf16 shadows, fp32 accumulate: tmjx_linear_nolds_bf16) — the behaviour log-prob stored with a
            # roll-out and the learner's first-pass log-prob then come from one set of numerics and the PPO ratio starts at 1
            bf = self.shadows if (lds_free and self.matmul_dtype == torch.bfloat16) else None

            def linear_bf16(a, sa_row, sa_k, lin, bias, mean=None, inv_std=None):
                wsh = bf.w[lin]
                Kp = (lin.in_features + 3) // 4 * 4          # (activation rows are padded to a multiple of 4 with zeros; the shadow is zero there too)
                out = torch.empty((n, lin.out_features), **f32)
                _hip.check(L.tmjx_linear_nolds_bf16(p(a), sa_row, sa_k, p(wsh), wsh.stride(0), p(lin.bias) if bias else None, p(out), n, lin.out_features,
                                                    Kp, p(mean), p(inv_std), stream), "tmjx_linear_nolds_bf16")
                return out

            # round 5: the layer through a 20 KB LDS tile (tmjx_linear_act) where the operands allow — the physics kernel's env image leaves that
            # much LDS free on every CU next to twelve resident envs; TMJX_ACT_LDS=0 keeps the LDS-free kernels (A/B runs)
            act_lds = os.environ.get("TMJX_ACT_LDS", "1") != "0"

            def linear_lds(a, sa_row, sa_k, w, bias_t, out, mean=None, inv_std=None):
                """True if the layer went through tmjx_linear_act (row-major a with aligned rows, K % 4 == 0)."""
                if not act_lds or sa_k != 1 or not L.tmjx_linear_act_ok(p(a), sa_row, p(w), w.stride(0), w.shape[1]):
                    return False
                _hip.check(L.tmjx_linear_act(p(a), sa_row, p(w), w.stride(0), p(bias_t), p(out), n, out.shape[1], w.shape[1], p(mean), p(inv_std), stream),
                           "tmjx_linear_act")
                return True

            def linear(a, sa_row, sa_k, K, lin, bias=True):
                if bf is not None and lin in bf.w:
                    return linear_bf16(a, sa_row, sa_k, lin, bias)
                # weights whose row length is not a multiple of 4 are used through a zero-padded copy (refreshed at the start of
                # every collect()): the kernel then takes its float4 path; the extra k read finite activations x 0
                w = self._padded_weight(lin)
                out = torch.empty((n, lin.out_features), **f32)
                if linear_lds(a, sa_row, sa_k, w, lin.bias if bias else None, out):
                    return out
                _hip.check(L.tmjx_linear_nolds(p(a), sa_row, sa_k, p(w), p(lin.bias) if bias else None, p(out),
                                               n, lin.out_features, w.shape[1], stream), "tmjx_linear_nolds")
                return out

            def block(a, sa_row, sa_k, K, blk, folded=False):
                if folded and bf is not None and blk.dense in bf.w:
                    z, bias_v = linear_bf16(a, sa_row, sa_k, blk.dense, False, self._fold[0], self._fold[1]), blk.dense.bias
                elif folded:      # the operand is normalised while it is loaded (mean / inv_std padded with 0 / 0 to the weight's padded K)
                    w = self._padded_weight(blk.dense)
                    z = torch.empty((n, blk.dense.out_features), **f32)
                    if not linear_lds(a, sa_row, sa_k, w, None, z, self._fold[0], self._fold[1]):
                        _hip.check(L.tmjx_linear_nolds_norm(p(a), sa_row, sa_k, p(w), None, p(z), n, blk.dense.out_features, w.shape[1],
                                                            p(self._fold[0]), p(self._fold[1]), stream), "tmjx_linear_nolds_norm")
                    bias_v = blk.dense.bias
                else:
                    z, bias_v = linear(a, sa_row, sa_k, K, blk.dense, bias=False), blk.dense.bias
                y = torch.empty_like(z)
                stats = torch.empty((n, 2), **f32)
                _hip.check(L.tmjx_silu_ln_fwd(p(z), p(bias_v), p(blk.norm.weight), p(blk.norm.bias), p(y), p(stats), n,
                                              blk.dense.out_features, float(blk.norm.eps), stream), "tmjx_silu_ln_fwd")
                return y

            # act_rng "device": the two noise arrays are drawn inside tmjx_latent_concat / tmjx_sample_action from a per-generator Philox
            # counter on the device (no torch generator in the inference graph, no normal_ launches); "torch": the learner's generator
            device_rng = self.act_rng == "device"
            if device_rng:
                rs = self._act_rng_state(gen)
                rng_state, rng_seed = rs[0], rs[1]
            eps = None if device_rng else torch.randn((n, Z), generator=gen, device=self.dev)
            if lds_free:
                h, first = None, True
                for blk in pol.encoder:
                    h = block(src, src.stride(0), src.stride(1), ref, blk, folded=fold) if first else block(h, h.shape[1], 1, h.shape[1], blk)
                    first = False
                fc2 = linear(h, h.shape[1], 1, h.shape[1], pol.fc2)
            else:
                fc2 = pol.fc2(pol.encoder(x[..., :ref]))
            wdec = Z + W - ref
            xdec = torch.empty((n, (wdec + 3) // 4 * 4 if lds_free else wdec), **f32)       # (the kernel zeroes the pad columns)
            _hip.check(L.tmjx_latent_concat(p(fc2), p(eps), p(src), p(xdec), n, Z, W, ref, src.stride(0), src.stride(1),
                                            p(self.normalizer.mean) if fold else None, p(self.normalizer.std) if fold else None,
                                            xdec.shape[1], rng_seed if device_rng else 0, p(rng_state) if device_rng else None, stream), "tmjx_latent_concat")
            if lds_free:
                h = xdec
                for blk in pol.decoder:
                    h = block(h, h.shape[1], 1, h.shape[1], blk)
                logits = linear(h, h.shape[1], 1, h.shape[1], pol.head)
            else:
                logits = pol.head(pol.decoder(xdec))
            noise = None if device_rng else torch.randn((n, A), generator=gen, device=self.dev)
            raw = torch.empty((n, A), **f32)
            action_t = torch.empty((A, n), **f32)
            logp = torch.empty(n, **f32)
            _hip.check(L.tmjx_sample_action(p(logits), p(noise), p(raw), p(action_t), p(logp), n, A, rng_seed if device_rng else 0,
                                            p(rng_state) if device_rng else None, stream), "tmjx_sample_action")
        mean, logvar = torch.chunk(fc2, 2, dim=-1)
        return action_t.t(), {"raw_action": raw, "log_prob": logp, "logits": logits, "latent_mean": mean, "latent_logvar": logvar}

    def _act_rng_state(self, gen):
        """(device counter [2] int64, Philox key, generator) of the acting noise stream that belongs to a torch generator."""
        rs = self._act_rng.get(id(gen))
        if rs is None:
            rs = self._act_rng[id(gen)] = (torch.zeros(2, dtype=torch.long, device=self.dev),
                                           (self._noise_seed ^ (0xD1B54A32D192ED03 * (len(self._act_rng) + 1))) & (2 ** 64 - 1), gen)
        return rs

    def _store_transition(self, env, st, extra, obs_dst0, obs_dst1, t: int, sl: slice, obs_dst2=None) -> None:
        import ctypes as C
        from .. import hip as _hip
        raw, logp = extra["raw_action"], extra["log_prob"]
        if not (raw.is_contiguous() and logp.is_contiguous() and st.obs.stride(0) == 1 and st.obs.stride(1) == st.obs.shape[0]):
            raise RuntimeError("roll-out store: unexpected layout of the policy outputs / the env's observation buffer")
        p = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
        b = self.buf
        q = _hip.RolloutStore(p(st.obs), p(obs_dst0), p(obs_dst1), p(raw), p(b["raw_action"][t, sl]), p(logp), p(b["log_prob"][t, sl]),
                              p(st.reward), p(b["reward"][t, sl]), p(st.done), p(b["discount"][t, sl]), p(st.info["truncation"]), p(b["truncation"][t, sl]),
                              st.obs.shape[0], st.obs.shape[1], raw.shape[1], p(obs_dst2))
        with torch.cuda.device(self.dev):
            _hip.check(_hip.lib().tmjx_rollout_store(C.byref(q), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)), "tmjx_rollout_store")

    def _padded_weight(self, lin) -> torch.Tensor:
        K = lin.in_features
        if K % 4 == 0:
            return lin.weight
        buf = self._wpad.get(lin)
        if buf is None:
            buf = torch.zeros((lin.out_features, (K + 3) // 4 * 4), dtype=torch.float32, device=self.dev)
            buf[:, :K].copy_(lin.weight.detach())
            self._wpad[lin] = buf
        return buf

    def _refresh_padded_weights(self) -> None:
        for lin, buf in self._wpad.items():
            buf[:, :lin.in_features].copy_(lin.weight.detach())
        if self.shadows is not None:
            self.shadows.refresh()       # bf16 mode: the acting policy reads the shadows too (the SGD step refreshes them only at its START)
        # mean and 1 / std of the reference part of the observation for tmjx_linear_nolds_norm (padded to the first layer's padded K: the pad
        # columns of the weight are zero); persistent buffers (the inference graphs hold their addresses), rewritten in place
        if self.lds_free and self.normalize_observations:
            K = self.policy.encoder[0].dense.in_features
            Kp = (K + 3) // 4 * 4
            with torch.no_grad():
                if getattr(self, "_fold", None) is None:
                    self._fold = (torch.zeros(Kp, dtype=torch.float32, device=self.dev), torch.zeros(Kp, dtype=torch.float32, device=self.dev))
                self._fold[0][:K].copy_(self.normalizer.mean[:K])
                torch.reciprocal(self.normalizer.std[:K], out=self._fold[1][:K])

    def _act_graphed(self, obs: torch.Tensor, g: int = 0):
        """act() replayed as one hipGraph per env group.  Valid while `obs` is the group's persistent observation buffer (same
        pointer every step); falls back to eager launches otherwise."""
        gen = self.gens[g]
        if not (self.use_graph and self.dev.type == "cuda") or os.environ.get("TMJX_NO_ACT_GRAPH"):
            return self.act(obs, gen=gen)
        key = (obs.data_ptr(), tuple(obs.shape), tuple(obs.stride()))
        ent = self._act_graphs.get(g)
        if ent is None or ent[1] != key:
            try:
                cur = torch.cuda.current_stream(self.dev)
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self.act(obs, gen=gen)
                cur.wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                graph.register_generator_state(gen)
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self.act(obs, gen=gen)
                ent = (graph, key, out)
            except Exception as e:  # noqa: BLE001
                print(f"[track_mjx_amd] hipGraph capture of the policy inference failed ({type(e).__name__}: {e}); running eagerly", flush=True)
                ent = (False, key, None)
                torch.cuda.synchronize(self.dev)
            self._act_graphs[g] = ent
        if ent[0] is False:
            return self.act(obs, gen=gen)
        ent[0].replay()
        return ent[2]

    @torch.no_grad()
    def collect(self) -> None:
        T, n_local = self.T, self.n_local
        offs = [0]
        for e in self.envs:
            offs.append(offs[-1] + e.num_envs)
        self._refresh_padded_weights()
        cur = torch.cuda.current_stream(self.dev) if self._streams else None
        if self._streams:
            for sg in self._streams:
                sg.wait_stream(cur)        # parameters / normaliser written by the previous update()
        jax_noise = self.sgd_keys is not None and self.act_rng == "jax"
        if jax_noise:
            self.sgd_keys.start_unrolls()
        for u in range(self.unrolls):
            if jax_noise:
                self.sgd_keys.start_unroll()
            for t in range(T):
                if jax_noise:      # the reference's own draws for all envs of this device, sliced per env group below (host-side: parity mode)
                    eps_np, noise_np = self.sgd_keys.act_noise(n_local, self.policy.latents, self.policy.action_size)
                    eps_all, noise_all = torch.from_numpy(eps_np).to(self.dev), torch.from_numpy(noise_np).to(self.dev)
                    if self._streams:      # uploaded on the current stream, consumed on the groups' streams: order them and keep the blocks alive there
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(self.dev))
                        for sg in self._streams:
                            sg.wait_event(ev)
                            eps_all.record_stream(sg); noise_all.record_stream(sg)
                for g, env in enumerate(self.envs):
                    sl = slice(u * n_local + offs[g], u * n_local + offs[g + 1])
                    with torch.cuda.stream(self._streams[g]) if self._streams else _nullctx():
                        st = self.states[g]
                        if u == 0 and t == 0:
                            self.buf["observation"][0, sl] = st.obs          # (every later row is written by the step that produces it)
                            if self._obs_rm is not None:
                                self._obs_rm[g].copy_(st.obs)
                        if jax_noise:
                            action, extra = self.act(st.obs, draws=(eps_all[offs[g]:offs[g + 1]], noise_all[offs[g]:offs[g + 1]]))
                        else:
                            action, extra = self._act_graphed(st.obs if self._obs_rm is None else self._obs_rm[g], g)
                        st = env.step(st, action)
                        # ONE launch stores this step's transition: the new observation (transposed into row t + 1 — or, at the end of an
                        # unroll, into next_observation_last and row 0 of the next unroll), raw action / log-prob of the acting policy
                        # (the inference graph's outputs are overwritten only by the next replay), reward, discount = 1 - done, truncation
                        nxt = self.buf["observation"][t + 1, sl] if t + 1 < T else self.buf["next_observation_last"][sl]
                        nxt1 = self.buf["observation"][0, slice(sl.start + n_local, sl.stop + n_local)] if (t + 1 == T and u + 1 < self.unrolls) else None
                        self._store_transition(env, st, extra, nxt, nxt1, t, sl, None if self._obs_rm is None else self._obs_rm[g])
                        self.states[g] = st
        if self._streams:
            for sg in self._streams:
                cur.wait_stream(sg)

    # ---- learning
    def _self_advancing(self) -> bool:
        """The SGD step draws its rows and its noise on the device (tmjx_minibatch_begin): needs the fused gather's layout."""
        return (self.dev.type == "cuda" and self.normalize_observations and self.buf["observation"].shape[-1] % 4 == 0
                and all(v.is_contiguous() for v in self.buf.values()) and not os.environ.get("TMJX_NO_SELF_ADVANCE"))

    def _mb_data(self, idx):
        fused_gather = self.dev.type == "cuda" and self.normalize_observations and self.buf["observation"].shape[-1] % 4 == 0
        if idx is None:
            data = _losses.minibatch_begin({**self.buf, "_B": self.local_batch}, self._perm_static, self._mb_state, self._noise_seed, self.normalizer,
                                           self.policy.latents, obs16=self._obs16)
        elif fused_gather and all(v.is_contiguous() for v in self.buf.values()):
            data = _losses.gather_minibatch(self.buf, idx, self.normalizer)      # all seven leaves in one launch
        else:
            data = {k: (self.buf[k].index_select(1, idx) if k != "next_observation_last" else self.buf[k].index_select(0, idx)) for k in self.buf}
        if self.shadows is not None:
            self.shadows.refresh()          # one launch: the optimiser step behind the previous replay changed the master weights
        return data

    def _mb_forward(self, idx, kl_w, scalars_on_side: bool = False):
        """GPU: gather + both networks' forward passes + the loss head (outside autograd).  Returns (network outputs, their gradients, the loss
        kernel's eight scalars) for _mb_backward."""
        data = self._mb_data(idx)
        twins = None
        if "observation_normalized_bf16" in data:
            on = data["observation_normalized"]
            twins = {on.data_ptr(): (data["observation_normalized_bf16"], on.shape[-1])}
        with gemm_inputs(self.matmul_dtype, self.shadows, twins):
            m, outs, gouts, out8 = _losses.ppo_loss_and_output_grads(self.policy, self.value, self.normalizer, data, kl_weight=kl_w,
                                                                     side_stream=self._sgd_side, acc_out=self._acc8 if idx is None else None,
                                                                     scalars_on_side=scalars_on_side and self._sgd_side is not None, **self.hp)
        return outs, gouts, out8

    def _mb_backward(self, outs, gouts, which: str = "all"):
        """One backward pass from the network outputs with the loss head's gradients as `grad_outputs`.  which = "all": both networks in one
        autograd call (the single-GPU step); "value" / "policy": one network — its gradients land in its bucket of the flat buffer, so that the
        bucket's all-reduce can start while the other network's pass still runs.  outs = (logits, baseline[, fc2])."""
        npol = self._n_policy_params
        if which == "all":
            o, g, params = list(outs), list(gouts), self.grads.params
        elif which == "value":
            o, g, params = [outs[1]], [gouts[1]], self.grads.params[npol:]
        else:
            o, g, params = [outs[0]] + list(outs[2:]), [gouts[0]] + list(gouts[2:]), self.grads.params[:npol]
        with gemm_inputs(self.matmul_dtype, self.shadows):
            with deferred_weight_grads() as dwg:
                grads = torch.autograd.grad(o, params, grad_outputs=g)
            if which != "policy" and self._sgd_side is not None:
                if which == "all" and self._value_dw_wgs > 0:
                    # the value network's weight gradients as a group of their own on the side stream, right behind that network's backward pass (autograd
                    # has queued it there): they run NEXT TO the policy's backward pass instead of behind it in the one launch at the end of the step
                    vptr = self._value_grad_ptrs
                    with torch.cuda.stream(self._sgd_side):
                        dwg.launch_subset(lambda gw: gw.data_ptr() in vptr, self._value_dw_wgs)
                torch.cuda.current_stream(self.dev).wait_stream(self._sgd_side)     # the value net's backward ran there
            dwg.launch()             # every (remaining) layer's (dW, db) in one grouped launch, straight into the flat gradient buffer
        setattr(self, "_dwg_" + which, dwg)          # (keeps the slab scratch alive until the next step)
        if which == "all":
            self.grads.assign(grads)
        else:
            self.grads.assign_subset(params, grads)

    def _minibatch_grads(self, idx: torch.Tensor | None, kl_w: float) -> torch.Tensor:
        """Gather one minibatch, loss, gradients into the flat buffer; returns the 5 loss terms as one tensor.  idx None: the self-advancing
        form — rows from the epoch's permutation at the device-side slot counter, noise from the device-side Philox stream, metrics added
        to self._acc8: a captured graph of it replays with NO host input (no index copy, no torch generator state to refresh)."""
        if self.dev.type == "cuda":
            # loss head outside autograd: its kernels give d loss / d(network outputs), one backward pass from the outputs
            outs, gouts, out8 = self._mb_forward(idx, kl_w, scalars_on_side=True)      # (_mb_backward("all") joins the side stream)
            self._mb_backward(outs, gouts, "all")
            if idx is None:
                return self._acc8                    # (the loss kernel added this step's scalars; reordered to METRIC_KEYS once per update())
            return out8[self._metric_index]          # (total, policy, v, kl, entropy) in METRIC_KEYS order: one gather
        data = self._mb_data(idx)
        with gemm_inputs(self.matmul_dtype, self.shadows):
            loss, m = _losses.compute_ppo_loss(self.policy, self.value, self.normalizer, data, kl_weight=kl_w, **self.hp,
                                               **({"gae_fn": self.gae_fn} if self.gae_fn is not None else {}))
        self.grads.assign(torch.autograd.grad(loss, self.grads.params))
        return torch.stack([m[k].float() for k in self.METRIC_KEYS])

    def _bucketed_step_eager(self, idx, kl_w):
        """The minibatch step with C1 in two buckets, launched eagerly (no hipGraph): forward + loss head, the value network's backward on the
        side stream followed by ITS bucket's all-reduce, the policy's backward on the main stream meanwhile, then the policy bucket.  Returns
        (metrics tensor, work handles to wait for in front of the optimiser)."""
        cur = torch.cuda.current_stream(self.dev)
        outs, gouts, out8 = self._mb_forward(idx, kl_w)
        side = self._c1_side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._mb_backward(outs, gouts, "value")
            works = self.grads.all_reduce_mean(self.group, force=True, lo=self._bucket_split, hi=None, async_op=True)
        self._mb_backward(outs, gouts, "policy")
        works += self.grads.all_reduce_mean(self.group, force=True, lo=0, hi=self._bucket_split, async_op=True)
        cur.wait_stream(side)
        return (self._acc8 if idx is None else out8[self._metric_index]), works

    def _capture_split(self, kl_w: float):
        """The bucketed step as THREE hipGraphs: G1 = gather + forward passes + loss head, Gv = the value network's backward (+ its weight
        gradients), Gp = the policy's.  update() replays G1, then Gv on a side stream and Gp on the main stream CONCURRENTLY (as the two
        branches of the single-GPU graph do), and issues each bucket's all-reduce eagerly behind its graph: the value bucket travels while Gp
        runs.  No collective is captured; every graph has its own memory pool (Gv and Gp replay side by side and must not share scratch)."""
        selfadv = self._self_advancing()
        self._g_idx = None if selfadv else torch.zeros(self.local_batch, dtype=torch.long, device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        draw0 = self._mb_state[0:1].clone()         # (as _capture: the warm-up runs leave the device-side draw counter where it was)
        warm = torch.cuda.Stream(device=self.dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            for _ in range(3):
                if selfadv:
                    self._mb_state[1:].zero_()
                outs, gouts, _ = self._mb_forward(self._g_idx, kl_w)
                self._mb_backward(outs, gouts, "value"); self._mb_backward(outs, gouts, "policy")
        cur.wait_stream(warm)
        self._mb_state[0:1].copy_(draw0)
        torch.cuda.synchronize(self.dev)
        g1, gv, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls events concurrently and must not invalidate the capture
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            outs, gouts, out8 = self._mb_forward(self._g_idx, kl_w)
            self._g_out = self._acc8 if selfadv else out8[self._metric_index]
        with torch.cuda.graph(gv, capture_error_mode="thread_local"):
            self._mb_backward(outs, gouts, "value")
        with torch.cuda.graph(gp, capture_error_mode="thread_local"):
            self._mb_backward(outs, gouts, "policy")
        del outs, gouts
        self._split_graphs, self._graph_kl, self._graph_selfadv = (g1, gv, gp), kl_w, selfadv

    def _bucketed_step_replay(self):
        g1, gv, gp = self._split_graphs
        cur, side = torch.cuda.current_stream(self.dev), self._c1_side
        g1.replay()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gv.replay()
            works = self.grads.all_reduce_mean(self.group, force=True, lo=self._bucket_split, hi=None, async_op=True)
        gp.replay()
        works += self.grads.all_reduce_mean(self.group, force=True, lo=0, hi=self._bucket_split, async_op=True)
        cur.wait_stream(side)
        return works

    METRIC_KEYS = ("total_loss", "policy_loss", "v_loss", "kl_latent_loss", "entropy_loss")
    # tests of the torch (CPU) branch only: a GAE implementation for hosts without the HIP kernel (the product default, tmjx_gae, raises there)
    gae_fn = None
    # minibatch shuffle (ppo.py:304-311: jax.random.permutation(key_perm, x), one permutation for every leaf): None = torch.randperm of the
    # learner's generator; otherwise a callable (update_index, rows) -> int64 permutation on the learner's device, e.g. the reference's own
    # draws from a jax key (jax_random.py: sgd_permutations)
    perm_fn = None

    def _capture(self, kl_w: float):
        """hipGraph of _minibatch_grads (torch.cuda.graphs): ~250 launches of the SGD step replayed as one graph launch.
        The minibatch row indices are a static device buffer; the optimiser and the gradient all-reduce stay eager, so the
        same graph serves any world size."""
        selfadv = self._self_advancing()
        self._g_idx = None if selfadv else torch.zeros(self.local_batch, dtype=torch.long, device=self.dev)
        draw0 = self._mb_state[0:1].clone()         # the warm-up runs draw noise: the capture leaves the device-side draw counter where it found it
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                if selfadv:
                    self._mb_state[1:].zero_()          # (warm-up runs read slot 0 of whatever the permutation buffer holds: valid rows)
                self._minibatch_grads(self._g_idx, kl_w)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        self._mb_state[0:1].copy_(draw0)
        graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls events concurrently (world > 1) and must not invalidate the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._g_out = self._minibatch_grads(self._g_idx, kl_w)
        self._graph, self._graph_kl, self._graph_selfadv = graph, kl_w, selfadv

    def update(self, it: int = 0, kl_schedule: Callable | None = None) -> dict:
        if self.normalize_observations:
            self.normalizer.update(self.buf["observation"], group=self.group, distributed=self.collectives)     # C2 (group None = default group)
        kl_w = kl_schedule(it) if kl_schedule is not None else self.kl_weight
        rows = self.buf["reward"].shape[1]
        use_graph = self.use_graph and self.dev.type == "cuda"
        bucketed = self.overlap_c1 and self.dev.type == "cuda"          # C1 in two buckets, overlapped with the policy's backward pass
        have = (self._split_graphs is not None) if bucketed else (self._graph is not None)
        selfadv = self._self_advancing()
        if use_graph and (not have or self._graph_kl != kl_w or self._graph_selfadv != selfadv):
            try:
                (self._capture_split if bucketed else self._capture)(kl_w)
            except Exception as e:  # noqa: BLE001 — capture is an optimisation: fall back to eager launches, loudly
                print(f"[track_mjx_amd] hipGraph capture of the SGD step failed ({type(e).__name__}: {e}); running eagerly", flush=True)
                self.use_graph = use_graph = False
                torch.cuda.synchronize(self.dev)
                # a capture that died between the loss head (which parks the KL term's fc2 gradient) and the policy's backward pass leaves that
                # gradient parked: drop it, or the first eager forward would raise "never consumed" instead of falling back
                h = getattr(self.policy, "latent_grad_handle", None)
                if h is not None:
                    h.take()
                    self.policy.latent_grad_handle = None
        acc = torch.zeros(len(self.METRIC_KEYS), dtype=torch.float32, device=self.dev)
        if selfadv:
            self._acc8.zero_()
        for upd in range(self.num_updates):
            # one permutation for every leaf (ppo.py:306-311)
            perm = torch.randperm(rows, generator=self.gen, device=self.dev) if self.perm_fn is None else self.perm_fn(upd, rows).to(self.dev)
            if selfadv:
                self._perm_static.copy_(perm)
                self._mb_state[1:2].zero_()                   # slot 0 of the new permutation
            for mb in range(self.num_minibatches):
                idx = None if selfadv else perm[mb * self.local_batch:(mb + 1) * self.local_batch]
                if use_graph and not selfadv:
                    self._g_idx.copy_(idx)
                if bucketed:
                    # C1 as two all-reduces, the value network's issued as soon as ITS backward pass is done (it overlaps the policy's)
                    if use_graph:
                        works, out = self._bucketed_step_replay(), self._g_out
                    else:
                        out, works = self._bucketed_step_eager(idx, kl_w)
                    for w in works:
                        w.wait()
                else:
                    if use_graph:
                        self._graph.replay()
                        out = self._g_out
                    else:
                        out = self._minibatch_grads(idx, kl_w)
                    if self.overlap_c1:     # CPU (gloo tests): the same two buckets, nothing to overlap with
                        self.grads.all_reduce_mean(self.group, force=True, lo=self._bucket_split, hi=None)
                        self.grads.all_reduce_mean(self.group, force=True, lo=0, hi=self._bucket_split)
                    else:
                        self.grads.all_reduce_mean(self.group, force=self.collectives)       # C1: one RCCL all-reduce per minibatch step
                self.opt.step()                               # clip_by_global_norm(10.0) -> adam (ppo.py:517-520), one fused launch
                if not selfadv:
                    acc += out
        if selfadv:
            acc = self._acc8[self._metric_index]
        acc = acc / (self.num_updates * self.num_minibatches)
        self._refresh_padded_weights()       # act() / the evaluator right after this update must not see last step's zero-padded copies
        res = {k: acc[i] for i, k in enumerate(self.METRIC_KEYS)}
        res["kl_weight"] = torch.as_tensor(kl_w)
        return res

    def start_epoch(self) -> None:
        """A new training epoch begins (ppo.py:729-730: fresh per-device keys); only matters with shuffle_rng="jax"."""
        if self.sgd_keys is not None:
            self.sgd_keys.start_epoch()

    def training_step(self, it: int = 0, kl_schedule=None) -> dict:
        if self.sgd_keys is not None:
            self.sgd_keys.start_training_step()
        self.collect()
        return self.update(it, kl_schedule)


def train(environment, num_timesteps: int, episode_length: int, ckpt_mgr=None, config_dict: dict | None = None, *,
          num_envs: int | None = None, num_evals: int = 1, num_resets_per_eval: int = 0, learning_rate: float = 1e-4,
          entropy_cost: float = 1e-4, discounting: float = 0.9, seed: int = 0, unroll_length: int = 10, batch_size: int = 32,
          num_minibatches: int = 16, num_updates_per_batch: int = 2, normalize_observations: bool = False, reward_scaling: float = 1.0,
          clipping_epsilon: float = 0.3, gae_lambda: float = 0.95, kl_weight: float = 1e-3, use_kl_schedule: bool = True,
          encoder_hidden_layer_sizes=(1024, 1024), decoder_hidden_layer_sizes=(1024, 1024), value_hidden_layer_sizes=(1024, 1024),
          intention_latent_size: int = 60, progress_fn: Callable[[int, dict], None] = lambda *a: None,
          max_training_steps: int | None = None, eval_env=None, num_eval_envs: int = 128, deterministic_eval: bool = False,
          matmul_dtype: torch.dtype | None = None, group=None, checkpoint_path: str | None = None, restore_from: str | None = None,
          shuffle_rng: str = "torch", act_rng: str = "device", action_repeat: int = 1,
          policy_params_fn: Callable[..., None] = lambda *args, **kwargs: None, checkpoint_callback: Callable[[int], None] | None = None, **unused):
    """ppo.train(environment, num_timesteps, episode_length, ...) -> (make_policy, params, metrics)  (ppo.py:128-172,809).

    `environment` is an un-wrapped MultiClipTracking holding THIS rank's envs; it is wrapped here exactly like
    ppo.py:469-475 (wrappers.wrap with the default use_lstm=True wrapper semantics).

    Checkpoints (reference: process 0 saves step 0 and, after every eval epoch, step `it`: ppo.py:700-711,787-795): `checkpoint_path` = a
    directory; rank 0 writes `<directory>/<it>/{policy.npz, train_state.npz, config/metadata}` there (agent/checkpoint.py:save_step_dir:
    the reference's Composite item names; normaliser, policy, value, Adam state, env_steps, noise-stream positions, config JSON),
    atomically, never over an existing step.  `ckpt_mgr` may be any object with `.directory` (an orbax CheckpointManager has one) or a
    path; orbax itself is not in this image.  `restore_from` = a step directory, a checkpoint directory (its latest step) or a .npz of
    save_npz: the whole training state comes back (checkpointing.load_training_state, ppo.py:561-567) — parameters, optimiser, env_steps,
    noise streams — and the run continues at the iteration after the restored one (the reference leaves its iteration restart as a TODO,
    ppo.py:670-677, and would then collide with the existing steps of the same directory).

    `policy_params_fn` (ppo.py:162,220-224): called by process 0 after every eval epoch exactly as ppo.py:762-781 does — keyword arguments
    `current_step` (the eval iteration), `jit_logging_inference_fn` (the DETERMINISTIC logging policy of ppo_networks.py:103-149:
    (params, observations, key_sample) -> (action, {"latent_mean", "latent_logvar"})), `params` ((normalizer, policy) state), a fresh
    `policy_params_fn_key` per call, and `render_video` = `it % config_dict["env_config"]["render_interval"] == 0` (every call when the
    config names no interval).  `checkpoint_callback(it)` follows every saved checkpoint (ppo.py:171,713-715; checkpointing.save)."""
    from ..environment import wrap
    # a list of environments = equal groups of this rank's envs whose roll-outs are pipelined on separate HIP streams (collect())
    env_list = [wrap(e, episode_length=int(episode_length), action_repeat=int(action_repeat)) for e in (environment if isinstance(environment, (list, tuple)) else [environment])]
    env = env_list[0]
    learner = PPOLearner(env_list if len(env_list) > 1 else env, encoder_layers=encoder_hidden_layer_sizes, decoder_layers=decoder_hidden_layer_sizes,
                         critic_layers=value_hidden_layer_sizes, latents=intention_latent_size, learning_rate=learning_rate,
                         entropy_cost=entropy_cost, discounting=discounting, reward_scaling=reward_scaling, gae_lambda=gae_lambda,
                         clipping_epsilon=clipping_epsilon, unroll_length=unroll_length, batch_size=batch_size,
                         num_minibatches=num_minibatches, num_updates_per_batch=num_updates_per_batch,
                         normalize_observations=normalize_observations, kl_weight=kl_weight, seed=seed, matmul_dtype=matmul_dtype, group=group,
                         shuffle_rng=shuffle_rng, act_rng=act_rng)
    from . import checkpoint as _ckpt
    if checkpoint_path is None and ckpt_mgr is not None:
        checkpoint_path = str(getattr(ckpt_mgr, "directory", ckpt_mgr))
    restored = {"env_steps": None, "iteration": None}
    if restore_from is not None:
        restored = _ckpt.restore(restore_from, learner)
    start_it = int(restored.get("iteration") or 0)

    def save_checkpoint(it: int, env_steps: int):
        if checkpoint_path is None or learner.rank != 0:
            return None
        out = _ckpt.save_step_dir(checkpoint_path, it, learner, config=config_dict, env_steps=env_steps)
        if checkpoint_callback is not None:
            try:        # ppo.py:713-717 and checkpointing.save: a failing user callback is logged, the training run goes on
                checkpoint_callback(it)
            except Exception as e:
                logging.warning(f"checkpoint callback failed at step {it}: {e}")
        return out
    env_step_per_training_step = learner.env_steps_per_training_step * int(action_repeat)      # ppo.py:260-262
    num_evals_after_init = max(num_evals - 1, 1)
    steps_per_epoch = int(math.ceil(num_timesteps / (num_evals_after_init * env_step_per_training_step * max(num_resets_per_eval, 1))))
    kl_schedule = _losses.create_ramp_schedule(max_value=kl_weight, ramp_steps=max(int(num_evals * 0.25), 1)) if use_kl_schedule else None
    reset_gen = torch.Generator().manual_seed(seed + 1 + learner.rank)

    def reset_all():
        for k, e in enumerate(env_list):
            learner.states[k] = e.reset(reset_gen)
    reset_all()
    # evaluator (ppo.py:629-668): an env of `num_eval_envs` wrapped like the training env; process 0 only (ppo.py:744)
    evaluator = None
    if eval_env is not None and learner.rank == 0:
        from .evaluator import Evaluator
        eval_gen = torch.Generator(device=learner.dev).manual_seed(seed * 1000 + 991)   # not the roll-out generators (registered with hipGraphs)
        evaluator = Evaluator(wrap(eval_env, episode_length=int(episode_length), action_repeat=int(action_repeat)),
                              lambda obs: learner.act(obs, deterministic=deterministic_eval, gen=eval_gen), episode_length=int(episode_length),
                              action_repeat=int(action_repeat), seed=seed + 7)
    render_interval = max(int(((config_dict or {}).get("env_config") or {}).get("render_interval", 1) or 1), 1)
    ppf_calls = [0]

    def current_policy_params():
        return ({k: v.clone() for k, v in learner.normalizer.state_dict().items()}, {k: v.clone() for k, v in learner.policy.state_dict().items()})

    logging_scratch: dict = {}          # ONE scratch copy of the policy module for the logging policy, made at its first call

    def _logging_key(key_sample):
        """A threefry key from whatever the caller holds: a jax-style uint32[2] key, an int seed, the (seed, call) tuple this train() hands to
        `policy_params_fn`, or None."""
        from .. import jax_random as _jr
        if key_sample is None:
            return _jr.PRNGKey(0)
        if isinstance(key_sample, (tuple, list)) and len(key_sample) == 2 and not hasattr(key_sample, "dtype"):
            return _jr.fold_in(_jr.PRNGKey(int(key_sample[0])), int(key_sample[1]))
        k = np.asarray(key_sample.cpu() if isinstance(key_sample, torch.Tensor) else key_sample)
        if k.shape == (2,):
            return k.astype(np.uint32)
        return _jr.PRNGKey(int(k))

    @torch.no_grad()
    def logging_inference_fn(params, observations, key_sample=None):
        """make_logging_inference_fn(ppo_network)(deterministic=True) (ppo_networks.py:103-149, jitted at ppo.py:514-515) with `params` =
        (normalizer state, policy state) — functionally: the live learner's networks are left as they were.  As in the reference, ONLY the
        action is deterministic: `policy_network.apply(*params, observations, key_network)` is called without `deterministic=True`
        (ppo_networks.py:117-119), so the latent is still SAMPLED — `key_sample -> split -> key_network` (:116), `-> split -> encoder_rng`
        (intention_network.py:104), `eps = normal(encoder_rng, logvar.shape)` (:85-88), drawn here by the pinned threefry restatement
        (track_mjx_amd/jax_random.py) — and the action is the mode of the distribution of the resulting logits (:124-130)."""
        import copy
        from .. import jax_random as _jr
        norm_sd, pol_sd = params[0], params[1]
        pol = logging_scratch.get("policy")
        if pol is None:
            pol = logging_scratch["policy"] = copy.deepcopy(learner.policy)
        pol.load_state_dict(pol_sd)
        obs = torch.as_tensor(observations, dtype=torch.float32, device=learner.dev)
        lead = obs.shape[:-1]
        obs = obs.reshape(-1, obs.shape[-1])
        if learner.normalize_observations:
            obs = (obs - norm_sd["mean"].to(learner.dev)) / norm_sd["std"].to(learner.dev)
        key_network = _jr.split(_logging_key(key_sample))[1]
        encoder_rng = _jr.split(key_network)[1]
        eps = torch.from_numpy(np.asarray(_jr.normal(encoder_rng, (*lead, pol.latents)), dtype=np.float32)).reshape(-1, pol.latents).to(learner.dev)
        logits, mean, logvar = pol(obs, eps=eps, deterministic=False)
        return (NormalTanh.mode(logits.float()).reshape(*lead, -1), {"latent_mean": mean.reshape(*lead, -1), "latent_logvar": logvar.reshape(*lead, -1)})
    metrics: dict = {}
    total_steps, done_steps = int(restored.get("env_steps") or 0), 0       # TrainingState.env_steps continues across a resume
    if restore_from is None:
        save_checkpoint(0, 0)                             # ppo.py:700-711: the initial parameters
    for it in range(start_it + 1, num_evals_after_init + 1):
        for _ in range(max(num_resets_per_eval, 1)):
            t0 = time.time()
            acc: dict = {}
            learner.start_epoch()
            for s in range(steps_per_epoch):
                m = learner.training_step(it, kl_schedule)
                for k, v in m.items():
                    acc[k] = acc.get(k, 0.0) + v
                done_steps += 1
                if max_training_steps is not None and done_steps >= max_training_steps:
                    break
            torch.cuda.synchronize(learner.dev)
            n_done = s + 1
            dt = time.time() - t0
            total_steps += n_done * env_step_per_training_step
            metrics = {"training/sps": n_done * env_step_per_training_step / dt, "training/walltime": dt,
                       **{f"training/{k}": float(v / n_done) for k, v in acc.items()}}
            if num_resets_per_eval > 0:
                reset_all()
            if max_training_steps is not None and done_steps >= max_training_steps:
                break
        if learner.rank == 0:
            if evaluator is not None:
                metrics = evaluator.run_evaluation(metrics)
            # ppo.py:759-781: the user's policy callback (rendering / logging in the reference's train.py:335-351) with the current parameters
            ppf_calls[0] += 1
            policy_params_fn(current_step=it, jit_logging_inference_fn=logging_inference_fn, params=current_policy_params(),
                             policy_params_fn_key=(int(seed), ppf_calls[0]), render_video=(it % render_interval == 0))
            progress_fn(total_steps, metrics)
        save_checkpoint(it, total_steps)                  # ppo.py:787-795: after every eval epoch, process 0, step = the iteration
        if max_training_steps is not None and done_steps >= max_training_steps:
            break

    def make_policy(params=None, deterministic: bool = False):
        """make_inference_fn(params) (ppo_networks.py:34-100).  `params` = None: the live learner; otherwise the (normalizer, policy[, value])
        state dicts returned by train(), or a checkpoint file — loaded into the learner's networks IN PLACE first."""
        if params is not None:
            if isinstance(params, (str, os.PathLike)):
                _ckpt.restore(params, learner, load_optimizer=False)
            else:
                norm_sd, pol_sd = params[0], params[1]
                with torch.no_grad():
                    for k, v in norm_sd.items():
                        getattr(learner.normalizer, k).copy_(v)
                    learner.policy.load_state_dict(pol_sd)     # copies into the existing (flat-buffer) parameters
                    if len(params) > 2 and params[2] is not None:
                        learner.value.load_state_dict(params[2])
                learner._refresh_padded_weights()
        return lambda obs, key=None: learner.act(obs, deterministic=deterministic)

    params = ({k: v.clone() for k, v in learner.normalizer.state_dict().items()}, {k: v.clone() for k, v in learner.policy.state_dict().items()},
              {k: v.clone() for k, v in learner.value.state_dict().items()})
    return make_policy, params, metrics
