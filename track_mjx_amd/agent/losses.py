"""PPO loss of the tracking learner.

Mirrors track_mjx/agent/mlp_ppo/losses.py:
  compute_gae          :39-100   -> HIP kernel `tmjx_gae` (include/tmjx.h), called under no_grad
  compute_ppo_loss     :103-245  clipped surrogate, value loss * 0.5 * 0.5, entropy bonus, AR(1)-prior latent KL
  create_ramp_schedule :248-290  linear KL-weight ramp (indexed by epoch)
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from .. import hip as _hip
from .networks import NormalTanh


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_: float = 1.0, discount: float = 0.99):
    """[T,B] fp32 device tensors -> (vs, advantages), both [T,B] (losses.py:39-100). Runs the HIP scan kernel."""
    T, B = rewards.shape
    args = [x.detach().contiguous().float() for x in (truncation, termination, rewards, values, bootstrap_value)]
    if not args[0].is_cuda:
        raise _hip.TmjxError("compute_gae runs on the GPU only (tmjx_gae); there is no CPU fallback")
    vs = torch.empty((T, B), dtype=torch.float32, device=rewards.device)
    adv = torch.empty_like(vs)
    L = _hip.lib()
    with torch.cuda.device(rewards.device):
        stream = C.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
        _hip.check(L.tmjx_gae(*[C.c_void_p(a.data_ptr()) for a in args], float(lambda_), float(discount),
                              C.c_void_p(vs.data_ptr()), C.c_void_p(adv.data_ptr()), T, B, stream), "tmjx_gae")
    return vs, adv


def gather_normalize(src: torch.Tensor, idx: torch.Tensor, normalizer) -> torch.Tensor:
    """(src[:, idx] - mean) / std in one kernel; src [T, R, W] (or [R, W]), idx int64 [B] on the GPU."""
    squeeze = src.dim() == 2
    if squeeze:
        src = src.unsqueeze(0)
    T, R, W = src.shape
    out = torch.empty((T, idx.shape[0], W), dtype=torch.float32, device=src.device)
    L = _hip.lib()
    with torch.cuda.device(src.device):
        stream = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
        _hip.check(L.tmjx_gather_normalize(*[C.c_void_p(t.data_ptr()) for t in (src, idx, normalizer.mean, normalizer.std, out)], T, R,
                                           idx.shape[0], W, stream), "tmjx_gather_normalize")
    return out[0] if squeeze else out


def gather_minibatch(buf: dict, idx: torch.Tensor, normalizer) -> dict:
    """Every leaf of the roll-out buffer at the minibatch rows `idx` (ppo.py:304-317), observations normalised, in ONE launch
    (tmjx_gather_minibatch).  Returns the dict compute_ppo_loss / ppo_loss_and_output_grads take."""
    obs, nxt, act = buf["observation"], buf["next_observation_last"], buf["raw_action"]
    T, R, W = obs.shape
    B, A = idx.shape[0], act.shape[-1]
    f32 = dict(dtype=torch.float32, device=obs.device)
    obs_n, next_n = torch.empty((T, B, W), **f32), torch.empty((B, W), **f32)
    act_g, sc = torch.empty((T, B, A), **f32), torch.empty((4, T, B), **f32)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(obs.device):
        _hip.check(_hip.lib().tmjx_gather_minibatch(p(obs), p(nxt), p(act), p(buf["log_prob"]), p(buf["reward"]), p(buf["discount"]), p(buf["truncation"]), p(idx),
                                                    p(normalizer.mean), p(normalizer.std), p(obs_n), p(next_n), p(act_g), p(sc), T, R, B, W, A,
                                                    C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)), "tmjx_gather_minibatch")
    return {"observation_normalized": obs_n, "next_observation_last_normalized": next_n, "raw_action": act_g, "log_prob": sc[0], "reward": sc[1],
            "discount": sc[2], "truncation": sc[3]}


def minibatch_begin(buf: dict, perm: torch.Tensor, state: torch.Tensor, seed: int, normalizer, latents: int, advance: bool = True,
                    obs16: torch.Tensor | None = None) -> dict:
    """First launch of a self-advancing SGD step (tmjx_minibatch_begin): the minibatch rows perm[slot B : slot B + B] (slot = state[1], B =
    perm.numel() // num_minibatches is implied by the caller through `buf["_B"]`) of every leaf, observations normalised, plus the step's two
    N(0, 1) arrays ("latent_eps" [T, B, latents], "entropy_noise" [T, B, A]) from the device-side Philox stream; advances state[0:2]."""
    obs, nxt, act = buf["observation"], buf["next_observation_last"], buf["raw_action"]
    T, R, W = obs.shape
    B, A = int(buf["_B"]), act.shape[-1]
    f32 = dict(dtype=torch.float32, device=obs.device)
    obs_n, next_n = torch.empty((T, B, W), **f32), torch.empty((B, W), **f32)
    act_g, sc = torch.empty((T, B, A), **f32), torch.empty((4, T, B), **f32)
    eps, noise = torch.empty((T, B, latents), **f32), torch.empty((T, B, A), **f32)
    d = lambda t: t.data_ptr()  # noqa: E731
    m = _hip.Minibatch(d(obs), d(nxt), d(act), d(buf["log_prob"]), d(buf["reward"]), d(buf["discount"]), d(buf["truncation"]), d(perm), d(normalizer.mean),
                       d(normalizer.std), d(obs_n), d(next_n), d(act_g), d(sc), d(eps), d(noise), d(state), seed & (2 ** 64 - 1), T, R, B, W, A, latents, int(advance))
    out = {"observation_normalized": obs_n, "next_observation_last_normalized": next_n, "raw_action": act_g, "log_prob": sc[0], "reward": sc[1],
           "discount": sc[2], "truncation": sc[3], "latent_eps": eps, "entropy_noise": noise}
    with torch.cuda.device(obs.device):
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        if obs16 is not None:
            # bf16 GEMM-input mode: the launch also writes the normalised observation as bf16 into the caller's [T B][ld >= W] buffer (zero
            # beyond W), which the first layers' GEMMs stage instead of the fp32 rows (networks.py: gemm_inputs.twins)
            assert obs16.dtype == torch.bfloat16 and obs16.shape[0] == T * B and obs16.shape[1] >= W and obs16.is_contiguous()
            _hip.check(_hip.lib().tmjx_minibatch_begin_bf16(C.byref(m), C.c_void_p(obs16.data_ptr()), obs16.shape[1], stream), "tmjx_minibatch_begin_bf16")
            out["observation_normalized_bf16"] = obs16
        else:
            _hip.check(_hip.lib().tmjx_minibatch_begin(C.byref(m), stream), "tmjx_minibatch_begin")
    return out


def create_ramp_schedule(max_value: float = 0.1, min_value: float = 0.0001, ramp_steps: int = 1000, warmup_steps: int = 0):
    """Linear ramp (losses.py:263-269): clip((step - warmup)/ramp_steps, min_value, 1) * max_value."""
    def schedule_fn(step: float) -> float:
        if step < warmup_steps:
            return min_value
        progress = min(max((step - warmup_steps) / ramp_steps, min_value), 1.0)
        return progress * max_value
    return schedule_fn


class _FusedLossHead(torch.autograd.Function):
    """tmjx_ppo_loss (csrc/ppo_kernels.h): the scalar loss and its gradients w.r.t. logits / baseline / fc2 in four
    launches; backward only scales the stored gradients by the incoming scalar."""

    @staticmethod
    def forward(ctx, logits, baseline, fc2, raw_action, behaviour_logp, noise, bootstrap, reward, discount, truncation, cfg):
        T, B = reward.shape
        dev = logits.device
        args = [a.detach().contiguous().float() for a in (logits, raw_action, behaviour_logp, noise, baseline, bootstrap, reward,
                                                          discount, truncation, fc2)]
        dlogits, dbaseline, dfc2 = torch.empty_like(args[0]), torch.empty_like(args[4]), torch.empty_like(args[9])
        L = _hip.lib()
        scratch = torch.empty(L.tmjx_ppo_scratch_floats(T, B), dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        c = _hip.PpoCfg(T, B, raw_action.shape[-1], fc2.shape[-1] // 2, cfg["reward_scaling"], cfg["discounting"], cfg["gae_lambda"],
                        cfg["clipping_epsilon"], cfg["entropy_cost"], cfg["kl_weight"], int(cfg["normalize_advantage"]), 0)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            ptr = [C.c_void_p(a.data_ptr()) for a in args + [dlogits, dbaseline, dfc2, scratch, out]]
            _hip.check(L.tmjx_ppo_loss(C.byref(c), *ptr, stream), "tmjx_ppo_loss")
        ctx.save_for_backward(dlogits, dbaseline, dfc2)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_out):
        dlogits, dbaseline, dfc2 = ctx.saved_tensors
        return (dlogits * g_total, dbaseline * g_total, dfc2 * g_total) + (None,) * 8


def ppo_loss_and_output_grads(policy, value, normalizer, data: dict, *, entropy_cost: float = 1e-4, kl_weight: float = 1e-3,
                              discounting: float = 0.9, reward_scaling: float = 1.0, gae_lambda: float = 0.95,
                              clipping_epsilon: float = 0.3, normalize_advantage: bool = True, side_stream=None, acc_out: torch.Tensor | None = None,
                              scalars_on_side: bool = False):
    """The learner's form of compute_ppo_loss_fused: network outputs with autograd, then tmjx_ppo_loss OUTSIDE autograd; returns
    (metrics, outputs, output_grads) so that the caller runs ONE torch.autograd.grad(outputs, params, grad_outputs=output_grads) —
    no autograd node for the loss head, no `grad * 1.0` passes over the three gradient arrays, no clone of the scalar."""
    obs = data["observation_normalized"] if "observation_normalized" in data else normalizer.normalize(data["observation"])
    nxt = data["next_observation_last_normalized"] if "next_observation_last_normalized" in data else normalizer.normalize(data["next_observation_last"])
    T, B = data["reward"].shape
    dev = obs.device
    L = _hip.lib()
    # the loss head in phases (tmjx_ppo_loss_phases) when the networks run on two streams: B — GAE, advantage statistics, value loss: the value network's
    # outputs only — goes onto the value network's stream, D (the eight scalars, which the backward pass does not wait for) leaves the main stream
    # too: the main stream runs policy forward, A, C, backward.  TMJX_PPO_PHASES=0: the one-call form
    phased = side_stream is not None and T <= 24 and os.environ.get("TMJX_PPO_PHASES", "1") != "0"
    cfg_of = lambda A_, Z_: _hip.PpoCfg(T, B, A_, Z_, reward_scaling, discounting, gae_lambda, clipping_epsilon, entropy_cost, kl_weight,  # noqa: E731
                                        int(normalize_advantage), int(acc_out is not None))
    f32 = lambda a: a.detach().contiguous().float()  # noqa: E731
    if side_stream is not None:
        # the value network is independent of the policy until the loss head: its forward — and, because autograd runs a node's
        # backward on the stream of its forward, its backward too — goes to a second stream (a parallel branch of the captured
        # graph): its GEMMs run next to the policy's epilogue / reduction kernels and vice versa
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            scratch = torch.empty(L.tmjx_ppo_scratch_floats(T, B), dtype=torch.float32, device=dev)
            out = torch.empty(8, dtype=torch.float32, device=dev) if acc_out is None else acc_out
            row = [f32(data[k]) for k in ("raw_action", "log_prob", "reward", "discount", "truncation")]
        side_stream.wait_stream(cur)
        with torch.cuda.stream(side_stream):
            # the bootstrap value FIRST: its 1 024-row launch is latency-bound (32 workgroups, 44 us) — queued behind the critic's 20 480-row pass it ran next to
            # the DECODER's pass at the end of the forward phase and held 32 CUs back from it (decoder chain 100 us against 78 alone); in front, it hides
            # under the two large passes' first round
            with torch.no_grad():
                bootstrap = value(nxt) if os.environ.get("TMJX_BOOTSTRAP_LAST") != "1" else None
            baseline = value(obs)
            with torch.no_grad():
                if bootstrap is None:
                    bootstrap = value(nxt)
                if phased:
                    bl, bs = f32(baseline), f32(bootstrap)
        logits, fc2 = policy(obs, eps=data.get("latent_eps"), return_fc2=True)
        if not phased:
            cur.wait_stream(side_stream)
        baseline.record_stream(cur); bootstrap.record_stream(cur)
    else:
        logits, fc2 = policy(obs, eps=data.get("latent_eps"), return_fc2=True)
        baseline = value(obs)
        with torch.no_grad():
            bootstrap = value(nxt)
    with torch.no_grad():
        noise = data["entropy_noise"] if "entropy_noise" in data else torch.randn(data["raw_action"].shape, dtype=torch.float32, device=logits.device)   # entropy sample (randn_like(loc))
        if phased:
            lg, nz, f2 = f32(logits), f32(noise), f32(fc2)
            dlogits, dbaseline, dfc2 = torch.empty_like(lg), torch.empty_like(bl), torch.empty_like(f2)
            c = cfg_of(data["raw_action"].shape[-1], fc2.shape[-1] // 2)
            args = [lg, row[0], row[1], nz, bl, bs, row[2], row[3], row[4], f2]
            ptr = [C.c_void_p(a.data_ptr()) for a in args + [dlogits, dbaseline, dfc2, scratch, out]]
            for t_ in (scratch, out, dbaseline, *row):
                t_.record_stream(side_stream)
            for t_ in (bl, bs):
                t_.record_stream(cur)

            def phase(mask, stream):
                _hip.check(L.tmjx_ppo_loss_phases(C.byref(c), *ptr, mask, C.c_void_p(stream.cuda_stream)), "tmjx_ppo_loss_phases")
            with torch.cuda.device(dev):
                phase(2, side_stream)           # B: behind the value network's forward pass, on its stream
                cur.wait_stream(side_stream)
                phase(1 | 4, cur)               # A and C behind the policy's forward pass, ONE launch (k_ppo_ac; C's half needs B's records)
                if scalars_on_side:             # (the caller joins `side_stream` into the current stream later: PPOLearner._mb_backward)
                    side_stream.wait_stream(cur)
                    phase(8, side_stream)       # D: the scalars — in front of the value network's backward pass on ITS stream, off the policy's path
                else:
                    phase(8, cur)
            args9 = f2
        else:
            args = [f32(a) for a in (logits, data["raw_action"], data["log_prob"], noise, baseline, bootstrap, data["reward"],
                                     data["discount"], data["truncation"], fc2)]
            dlogits, dbaseline, dfc2 = torch.empty_like(args[0]), torch.empty_like(args[4]), torch.empty_like(args[9])
            if side_stream is None:
                scratch = torch.empty(L.tmjx_ppo_scratch_floats(T, B), dtype=torch.float32, device=dev)
                # `acc_out`: the caller's running sums of the eight loss scalars (the kernel ADDS to them: no add launch per minibatch step)
                out = torch.empty(8, dtype=torch.float32, device=dev) if acc_out is None else acc_out
            c = cfg_of(data["raw_action"].shape[-1], fc2.shape[-1] // 2)
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                ptr = [C.c_void_p(a.data_ptr()) for a in args + [dlogits, dbaseline, dfc2, scratch, out]]
                _hip.check(L.tmjx_ppo_loss(C.byref(c), *ptr, stream), "tmjx_ppo_loss")
            args9 = args[9]
    metrics = {"total_loss": out[0], "policy_loss": out[1], "v_loss": out[2], "kl_latent_loss": out[4], "entropy_loss": out[3]}
    handle = getattr(policy, "latent_grad_handle", None)
    if handle is not None and handle.matches(fc2) and args9.data_ptr() == fc2.data_ptr():
        # fc2 has two gradients, the KL term's (here) and the latent sample's (tmjx_latent_concat_bwd): hand this one to that kernel, which sums
        # them, instead of seeding autograd with it (autograd would add the two with an element-wise launch of its own; same sum, same bits).
        # It is parked on THIS forward pass's handle: a backward pass that does not run through this forward leaves it there, and the policy's
        # next forward raises instead of dropping it
        handle.add(dfc2.view(-1, fc2.shape[-1]))
        return metrics, (logits, baseline), (dlogits.view_as(logits), dbaseline.view_as(baseline)), out
    return metrics, (logits, baseline, fc2), (dlogits.view_as(logits), dbaseline.view_as(baseline), dfc2.view_as(fc2)), out


def compute_ppo_loss_fused(policy, value, normalizer, data: dict, *, entropy_cost: float = 1e-4, kl_weight: float = 1e-3,
                           discounting: float = 0.9, reward_scaling: float = 1.0, gae_lambda: float = 0.95,
                           clipping_epsilon: float = 0.3, normalize_advantage: bool = True):
    """compute_ppo_loss with the loss head (everything after the network outputs) in the HIP kernels of
    csrc/ppo_kernels.h.  Same inputs, same outputs; GPU only (the product path of PPOLearner.update)."""
    # `*_normalized` entries: already gathered + normalised by tmjx_gather_normalize (PPOLearner._minibatch_grads)
    obs = data["observation_normalized"] if "observation_normalized" in data else normalizer.normalize(data["observation"])
    logits, fc2 = policy(obs, return_fc2=True)
    baseline = value(obs)
    with torch.no_grad():
        nxt = data["next_observation_last_normalized"] if "next_observation_last_normalized" in data else normalizer.normalize(data["next_observation_last"])
        bootstrap_value = value(nxt)
        noise = torch.randn(data["raw_action"].shape, dtype=torch.float32, device=logits.device)   # entropy sample (randn_like(loc))
    cfg = dict(reward_scaling=reward_scaling, discounting=discounting, gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon,
               entropy_cost=entropy_cost, kl_weight=kl_weight, normalize_advantage=normalize_advantage)
    total, out = _FusedLossHead.apply(logits, baseline, fc2, data["raw_action"], data["log_prob"], noise, bootstrap_value, data["reward"],
                                      data["discount"], data["truncation"], cfg)
    return total, {"total_loss": out[0], "policy_loss": out[1], "v_loss": out[2], "kl_latent_loss": out[4], "entropy_loss": out[3],
                   "kl_weight": torch.as_tensor(kl_weight)}


def compute_ppo_loss(policy, value, normalizer, data: dict, *, entropy_cost: float = 1e-4, kl_weight: float = 1e-3,
                     discounting: float = 0.9, reward_scaling: float = 1.0, gae_lambda: float = 0.95,
                     clipping_epsilon: float = 0.3, normalize_advantage: bool = True, gae_fn=compute_gae):
    """data (time-major): observation [T,B,obs], next_observation_last [B,obs], reward/discount/truncation/log_prob [T,B],
    raw_action [T,B,nu].  Returns (total_loss, metrics) exactly as losses.py:103-245."""
    obs = normalizer.normalize(data["observation"])
    logits, latent_mean, latent_logvar = policy(obs)
    baseline = value(obs)
    with torch.no_grad():
        bootstrap_value = value(normalizer.normalize(data["next_observation_last"]))
    rewards = data["reward"] * reward_scaling
    truncation = data["truncation"]
    termination = (1 - data["discount"]) * (1 - truncation)
    target_log_probs = NormalTanh.log_prob(logits, data["raw_action"])
    behaviour_log_probs = data["log_prob"]
    vs, advantages = gae_fn(truncation, termination, rewards, baseline.detach(), bootstrap_value, gae_lambda, discounting)
    if normalize_advantage:
        advantages = (advantages - advantages.mean()) / (advantages.std(unbiased=False) + 1e-8)
    rho = torch.exp(target_log_probs - behaviour_log_probs)
    policy_loss = -torch.mean(torch.minimum(rho * advantages, torch.clamp(rho, 1 - clipping_epsilon, 1 + clipping_epsilon) * advantages))
    v_error = vs - baseline
    v_loss = torch.mean(v_error * v_error) * 0.5 * 0.5
    entropy = torch.mean(NormalTanh.entropy(logits))
    entropy_loss = entropy_cost * -entropy
    # latent KL: t = 0 against N(0, I); t >= 1 against the AR(1) prior N(0.95 mu_{t-1}, (1 - 0.95^2) I)
    alpha = 0.95
    prior_variance = 1 - alpha ** 2
    kl_0 = -0.5 * torch.mean(1 + latent_logvar[0] - latent_mean[0] ** 2 - torch.exp(latent_logvar[0]))
    T = latent_mean.shape[0]
    if T > 1:
        z_prev, mu_curr, logvar_curr = latent_mean[:-1], latent_mean[1:], latent_logvar[1:]
        var_ratio = torch.exp(logvar_curr) / prior_variance
        mean_diff_sq = (alpha * z_prev - mu_curr) ** 2 / prior_variance
        log_var_ratio = torch.log(torch.tensor(prior_variance, device=obs.device)) - logvar_curr
        kl_t = 0.5 * torch.mean(var_ratio + mean_diff_sq - 1 + log_var_ratio)
        kl_latent_loss = kl_weight * ((kl_0 + kl_t * (T - 1)) / T)
    else:
        kl_latent_loss = kl_weight * kl_0
    total = policy_loss + v_loss + entropy_loss + kl_latent_loss
    return total, {"total_loss": total.detach(), "policy_loss": policy_loss.detach(), "v_loss": v_loss.detach(),
                   "kl_latent_loss": kl_latent_loss.detach(), "entropy_loss": entropy_loss.detach(),
                   "kl_weight": torch.as_tensor(kl_weight)}
