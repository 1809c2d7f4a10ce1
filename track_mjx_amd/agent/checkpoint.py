"""Parameter export / import in the reference's flax naming (first half).

The reference checkpoints `(normalizer_params, policy_params)` with orbax (track_mjx/agent/checkpointing.py:165-198); the
policy's flax tree is {'params': {'encoder': {hidden_i, LayerNorm_i, fc2_mean, fc2_logvar}, 'decoder': {hidden_i, LayerNorm_i}}}
(agent/mlp_ppo/intention_network.py:32-44,68-76: Dense kernels are [in, out], LayerNorm has scale / bias, the decoder's last
`hidden_L` is the un-activated output layer) and the value net is brax's MLP {'params': {hidden_i}}.  orbax / tensorstore are not in
this image, so the container here is a flat .npz whose keys are the '/'-joined tree paths — the tree a reference-side loader needs
to hand to orbax, three lines with `flax.traverse_util.unflatten_dict`.  `from_flax_tree` loads such a tree back (e.g. a
reference checkpoint converted the other way), so policies can move in both directions.
"""
from __future__ import annotations

import numpy as np
import torch

from .networks import IntentionPolicy, RunningStatistics, ValueNet


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().copy()


def policy_to_flax(policy: IntentionPolicy) -> dict:
    enc, dec = {}, {}
    for i, blk in enumerate(policy.encoder):
        enc[f"hidden_{i}"] = {"kernel": _np(blk.dense.weight.t()), "bias": _np(blk.dense.bias)}
        enc[f"LayerNorm_{i}"] = {"scale": _np(blk.norm.weight), "bias": _np(blk.norm.bias)}
    Z = policy.latents
    w, b = policy.fc2.weight, policy.fc2.bias            # the two heads are the halves of one GEMM here
    enc["fc2_mean"] = {"kernel": _np(w[:Z].t()), "bias": _np(b[:Z])}
    enc["fc2_logvar"] = {"kernel": _np(w[Z:].t()), "bias": _np(b[Z:])}
    for i, blk in enumerate(policy.decoder):
        dec[f"hidden_{i}"] = {"kernel": _np(blk.dense.weight.t()), "bias": _np(blk.dense.bias)}
        dec[f"LayerNorm_{i}"] = {"scale": _np(blk.norm.weight), "bias": _np(blk.norm.bias)}
    dec[f"hidden_{len(policy.decoder)}"] = {"kernel": _np(policy.head.weight.t()), "bias": _np(policy.head.bias)}
    return {"params": {"encoder": enc, "decoder": dec}}


def value_to_flax(value: ValueNet) -> dict:
    dense = [m for m in value.net if isinstance(m, torch.nn.Linear)]
    return {"params": {f"hidden_{i}": {"kernel": _np(l.weight.t()), "bias": _np(l.bias)} for i, l in enumerate(dense)}}


def normalizer_to_flax(n: RunningStatistics) -> dict:
    """brax RunningStatisticsState fields (count, mean, summed_variance, std)."""
    return {"count": _np(n.count), "mean": _np(n.mean), "summed_variance": _np(n.summed_variance), "std": _np(n.std)}


@torch.no_grad()
def policy_from_flax(policy: IntentionPolicy, tree: dict) -> None:
    p = tree["params"]
    dev = policy.fc2.weight.device

    def put(dst: torch.Tensor, src) -> None:
        dst.copy_(torch.as_tensor(np.asarray(src), dtype=dst.dtype, device=dev))

    for name, blocks in (("encoder", policy.encoder), ("decoder", policy.decoder)):
        for i, blk in enumerate(blocks):
            put(blk.dense.weight, np.asarray(p[name][f"hidden_{i}"]["kernel"]).T)
            put(blk.dense.bias, p[name][f"hidden_{i}"]["bias"])
            put(blk.norm.weight, p[name][f"LayerNorm_{i}"]["scale"])
            put(blk.norm.bias, p[name][f"LayerNorm_{i}"]["bias"])
    Z = policy.latents
    put(policy.fc2.weight[:Z], np.asarray(p["encoder"]["fc2_mean"]["kernel"]).T)
    put(policy.fc2.weight[Z:], np.asarray(p["encoder"]["fc2_logvar"]["kernel"]).T)
    put(policy.fc2.bias[:Z], p["encoder"]["fc2_mean"]["bias"])
    put(policy.fc2.bias[Z:], p["encoder"]["fc2_logvar"]["bias"])
    last = p["decoder"][f"hidden_{len(policy.decoder)}"]
    put(policy.head.weight, np.asarray(last["kernel"]).T)
    put(policy.head.bias, last["bias"])


@torch.no_grad()
def normalizer_from_flax(n: RunningStatistics, tree: dict) -> None:
    for k in ("count", "mean", "summed_variance", "std"):
        getattr(n, k).copy_(torch.as_tensor(np.asarray(tree[k]), dtype=torch.float32, device=n.mean.device).reshape(getattr(n, k).shape))


@torch.no_grad()
def value_from_flax(value: ValueNet, tree: dict) -> None:
    """brax value MLP {'params': {hidden_i: {kernel [in, out], bias}}} -> the Linear layers of ValueNet, in place (flat-buffer views
    and hipGraph pointers of a live learner stay valid)."""
    p = tree["params"]
    dense = [m for m in value.net if isinstance(m, torch.nn.Linear)]
    if len(dense) != len(p):
        raise ValueError(f"value tree has {len(p)} layers, the network {len(dense)}")
    for i, lin in enumerate(dense):
        lin.weight.copy_(torch.as_tensor(np.asarray(p[f"hidden_{i}"]["kernel"]).T.copy(), dtype=lin.weight.dtype, device=lin.weight.device))
        lin.bias.copy_(torch.as_tensor(np.asarray(p[f"hidden_{i}"]["bias"]), dtype=lin.bias.dtype, device=lin.bias.device))


def flatten(tree: dict, prefix: str = "") -> dict:
    out = {}
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, key))
        else:
            out[key] = np.asarray(v)
    return out


def unflatten(flat: dict) -> dict:
    tree: dict = {}
    for key, v in flat.items():
        node = tree
        parts = key.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = np.asarray(v)
    return tree


def learner_tree(learner) -> dict:
    """Everything a resumed run needs, in the reference's tree naming: (normalizer, policy, value) as the reference checkpoints them
    (checkpointing.py:280-299: `policy` = (normalizer_params, policy_params), `train_state` also holds the value params and the
    optimizer state) plus the Adam moments / step count of the flat optimiser (optax ScaleByAdamState: count, mu, nu)."""
    opt = learner.opt
    return {"normalizer": normalizer_to_flax(learner.normalizer), "policy": policy_to_flax(learner.policy), "value": value_to_flax(learner.value),
            "optimizer": {"count": np.asarray(opt.t, dtype=np.int64), "mu": _np(opt.exp_avg), "nu": _np(opt.exp_avg_sq)}}


def save_npz(path, learner, config: dict | None = None, step: int | None = None) -> None:
    """One flat .npz: learner_tree() + optionally the run's config as JSON (the reference embeds it: checkpointing.py:292-296) and
    the env-step counter."""
    import json
    flat = flatten(learner_tree(learner))
    if config is not None:
        flat["config_json"] = np.frombuffer(json.dumps(config, default=str).encode(), dtype=np.uint8)
    if step is not None:
        flat["env_steps"] = np.asarray(step, dtype=np.int64)
    np.savez(path, **flat)


def load_npz(path, learner, load_optimizer: bool = True) -> dict:
    """Restore normaliser, policy, value and (if present) the optimiser moments of a learner IN PLACE; returns {config, env_steps}."""
    import json
    with np.load(path) as z:
        flat = {k: z[k] for k in z.files}
    extra = {"config": json.loads(bytes(flat.pop("config_json")).decode()) if "config_json" in flat else None,
             "env_steps": int(flat.pop("env_steps")) if "env_steps" in flat else None}
    tree = unflatten(flat)
    normalizer_from_flax(learner.normalizer, tree["normalizer"])
    policy_from_flax(learner.policy, tree["policy"])
    if "value" in tree:
        value_from_flax(learner.value, tree["value"])
    if load_optimizer and "optimizer" in tree:
        o = tree["optimizer"]
        with torch.no_grad():
            learner.opt.exp_avg.copy_(torch.as_tensor(o["mu"], device=learner.opt.exp_avg.device))
            learner.opt.exp_avg_sq.copy_(torch.as_tensor(o["nu"], device=learner.opt.exp_avg_sq.device))
        learner.opt.t = int(o["count"])
    if hasattr(learner, "_refresh_padded_weights"):
        learner._refresh_padded_weights()
    return extra
