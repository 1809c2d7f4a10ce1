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


def _ident(p):
    return p


def policy_to_flax(policy: IntentionPolicy, get=_ident) -> dict:
    """`get(parameter) -> tensor of the same shape`: identity = the parameter values; the optimiser export passes the parameter's view
    of a moment buffer, so mu / nu come out as trees with the parameters' own names (optax ScaleByAdamState.mu / .nu)."""
    enc, dec = {}, {}
    for i, blk in enumerate(policy.encoder):
        enc[f"hidden_{i}"] = {"kernel": _np(get(blk.dense.weight).t()), "bias": _np(get(blk.dense.bias))}
        enc[f"LayerNorm_{i}"] = {"scale": _np(get(blk.norm.weight)), "bias": _np(get(blk.norm.bias))}
    Z = policy.latents
    w, b = get(policy.fc2.weight), get(policy.fc2.bias)            # the two heads are the halves of one GEMM here
    enc["fc2_mean"] = {"kernel": _np(w[:Z].t()), "bias": _np(b[:Z])}
    enc["fc2_logvar"] = {"kernel": _np(w[Z:].t()), "bias": _np(b[Z:])}
    for i, blk in enumerate(policy.decoder):
        dec[f"hidden_{i}"] = {"kernel": _np(get(blk.dense.weight).t()), "bias": _np(get(blk.dense.bias))}
        dec[f"LayerNorm_{i}"] = {"scale": _np(get(blk.norm.weight)), "bias": _np(get(blk.norm.bias))}
    dec[f"hidden_{len(policy.decoder)}"] = {"kernel": _np(get(policy.head.weight).t()), "bias": _np(get(policy.head.bias))}
    return {"params": {"encoder": enc, "decoder": dec}}


def value_to_flax(value: ValueNet, get=_ident) -> dict:
    dense = [m for m in value.net if isinstance(m, torch.nn.Linear)]
    return {"params": {f"hidden_{i}": {"kernel": _np(get(l.weight).t()), "bias": _np(get(l.bias))} for i, l in enumerate(dense)}}


def normalizer_to_flax(n: RunningStatistics) -> dict:
    """brax RunningStatisticsState fields (count, mean, summed_variance, std)."""
    return {"count": _np(n.count), "mean": _np(n.mean), "summed_variance": _np(n.summed_variance), "std": _np(n.std)}


@torch.no_grad()
def policy_from_flax(policy: IntentionPolicy, tree: dict, get=_ident) -> None:
    """In place.  `get(parameter)` = the destination tensor (identity: the parameter; the optimiser import passes moment-buffer views)."""
    p = tree["params"]
    dev = policy.fc2.weight.device

    def put(dst: torch.Tensor, src) -> None:
        dst.copy_(torch.as_tensor(np.ascontiguousarray(src), dtype=dst.dtype, device=dev))

    for name, blocks in (("encoder", policy.encoder), ("decoder", policy.decoder)):
        for i, blk in enumerate(blocks):
            put(get(blk.dense.weight), np.asarray(p[name][f"hidden_{i}"]["kernel"]).T)
            put(get(blk.dense.bias), p[name][f"hidden_{i}"]["bias"])
            put(get(blk.norm.weight), p[name][f"LayerNorm_{i}"]["scale"])
            put(get(blk.norm.bias), p[name][f"LayerNorm_{i}"]["bias"])
    Z = policy.latents
    w, b = get(policy.fc2.weight), get(policy.fc2.bias)
    put(w[:Z], np.asarray(p["encoder"]["fc2_mean"]["kernel"]).T)
    put(w[Z:], np.asarray(p["encoder"]["fc2_logvar"]["kernel"]).T)
    put(b[:Z], p["encoder"]["fc2_mean"]["bias"])
    put(b[Z:], p["encoder"]["fc2_logvar"]["bias"])
    last = p["decoder"][f"hidden_{len(policy.decoder)}"]
    put(get(policy.head.weight), np.asarray(last["kernel"]).T)
    put(get(policy.head.bias), last["bias"])


@torch.no_grad()
def normalizer_from_flax(n: RunningStatistics, tree: dict) -> None:
    for k in ("count", "mean", "summed_variance", "std"):
        getattr(n, k).copy_(torch.as_tensor(np.asarray(tree[k]), dtype=torch.float32, device=n.mean.device).reshape(getattr(n, k).shape))


@torch.no_grad()
def value_from_flax(value: ValueNet, tree: dict, get=_ident) -> None:
    """brax value MLP {'params': {hidden_i: {kernel [in, out], bias}}} -> the Linear layers of ValueNet, in place (flat-buffer views
    and hipGraph pointers of a live learner stay valid)."""
    p = tree["params"]
    dense = [m for m in value.net if isinstance(m, torch.nn.Linear)]
    if len(dense) != len(p):
        raise ValueError(f"value tree has {len(p)} layers, the network {len(dense)}")
    for i, lin in enumerate(dense):
        get(lin.weight).copy_(torch.as_tensor(np.asarray(p[f"hidden_{i}"]["kernel"]).T.copy(), dtype=lin.weight.dtype, device=lin.weight.device))
        get(lin.bias).copy_(torch.as_tensor(np.asarray(p[f"hidden_{i}"]["bias"]), dtype=lin.bias.dtype, device=lin.bias.device))


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


def _moment_view(learner, buf: torch.Tensor):
    """parameter -> its (un-padded) view of a flat moment buffer: the export is independent of the flat layout's pad rule."""
    from .ppo import _flat_view
    segs = {id(p): seg for p, seg in zip(learner.grads.params, learner.grads.segs)}
    return lambda p: _flat_view(buf, segs[id(p)], p)


def rng_tree(learner) -> dict:
    """The noise-stream positions a resumed run continues from: the SGD step's device-side Philox draw counter (`_mb_state[0]`), the
    acting streams' counters (one per env group, in creation order) and the torch generators' states (shuffles, the non-default noise)."""
    out = {"sgd_draw_counter": _np(learner._mb_state[:1]), "rank": np.asarray(int(getattr(learner, "rank", 0)), dtype=np.int64),
           "torch_generators": {str(i): g.get_state().cpu().numpy().copy() for i, g in enumerate(learner.gens)}}
    by_gen = {id(g): i for i, g in enumerate(learner.gens)}
    out["act_counters"] = {str(by_gen[k]): _np(v[0]) for k, v in learner._act_rng.items() if k in by_gen}
    return out


@torch.no_grad()
def rng_from_tree(learner, tree: dict) -> None:
    learner._mb_state[:1].copy_(torch.as_tensor(np.asarray(tree["sgd_draw_counter"]), device=learner._mb_state.device))
    # the generator states are the SAVING rank's (rank 0 writes the checkpoint): that rank continues its streams exactly; every other rank
    # derives its own continuation from (saved state, rank) — restoring rank 0's states everywhere would make all ranks draw identical
    # shuffles / noise from here on (the learner seeds them seed * 1000 + 17 + rank).  The device-side Philox counters below are shared on
    # purpose: their KEYS are per rank
    saved_rank, rank = int(np.asarray(tree.get("rank", 0))), int(getattr(learner, "rank", 0))
    for i, st in tree.get("torch_generators", {}).items():
        if int(i) < len(learner.gens):
            st = np.asarray(st).astype(np.uint8)
            if rank == saved_rank:
                learner.gens[int(i)].set_state(torch.as_tensor(st, dtype=torch.uint8))
            else:
                import hashlib
                h = int.from_bytes(hashlib.blake2b(st.tobytes() + rank.to_bytes(4, "little") + int(i).to_bytes(4, "little"), digest_size=8).digest(), "little")
                learner.gens[int(i)].manual_seed(h & (2 ** 63 - 1))
    for i, v in tree.get("act_counters", {}).items():
        if int(i) < len(learner.gens):
            rs = learner._act_rng_state(learner.gens[int(i)])
            rs[0].copy_(torch.as_tensor(np.asarray(v).astype(np.int64), device=rs[0].device))


def learner_tree(learner) -> dict:
    """Everything a resumed run needs, in the reference's tree naming: (normalizer, policy, value) as the reference checkpoints them
    (checkpointing.py:280-299: `policy` = (normalizer_params, policy_params), `train_state` also holds the value params and the
    optimizer state) plus the Adam state as optax's ScaleByAdamState (count, mu, nu; mu / nu are trees named like the parameters,
    un-padded) and the noise-stream positions."""
    opt = learner.opt
    mu, nu = _moment_view(learner, opt.exp_avg), _moment_view(learner, opt.exp_avg_sq)
    return {"normalizer": normalizer_to_flax(learner.normalizer), "policy": policy_to_flax(learner.policy), "value": value_to_flax(learner.value),
            "optimizer": {"count": np.asarray(opt.t, dtype=np.int64),
                          "mu": {"policy": policy_to_flax(learner.policy, mu), "value": value_to_flax(learner.value, mu)},
                          "nu": {"policy": policy_to_flax(learner.policy, nu), "value": value_to_flax(learner.value, nu)}},
            "rng": rng_tree(learner)}


def _atomic_savez(path, flat: dict, overwrite: bool) -> None:
    """np.savez to a temporary file in the same directory, then os.replace: a crash mid-write never leaves a truncated checkpoint under
    the final name.  An existing file is refused unless `overwrite` (orbax refuses to save an existing step)."""
    import os
    import tempfile
    path = str(path)
    if os.path.exists(path) and not overwrite:
        raise FileExistsError(f"{path} exists: a checkpoint is never overwritten (resume into a new directory, or pass overwrite=True)")
    fd, tmp = tempfile.mkstemp(prefix=".tmp-", suffix=".npz", dir=os.path.dirname(path) or ".")
    try:
        with os.fdopen(fd, "wb") as f:
            np.savez(f, **flat)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


def save_npz(path, learner, config: dict | None = None, step: int | None = None, iteration: int | None = None, overwrite: bool = False) -> None:
    """One flat .npz: learner_tree() + optionally the run's config as JSON (the reference embeds it: checkpointing.py:292-296), the
    env-step counter (TrainingState.env_steps) and the eval iteration (the reference's checkpoint step, ppo.py:787-795)."""
    import json
    flat = flatten(learner_tree(learner))
    if config is not None:
        flat["config_json"] = np.frombuffer(json.dumps(config, default=str).encode(), dtype=np.uint8)
    if step is not None:
        flat["env_steps"] = np.asarray(step, dtype=np.int64)
    if iteration is not None:
        flat["iteration"] = np.asarray(iteration, dtype=np.int64)
    _atomic_savez(path, flat, overwrite)


def load_npz(path, learner, load_optimizer: bool = True) -> dict:
    """Restore normaliser, policy, value and (if present) the optimiser moments and noise-stream positions of a learner IN PLACE;
    returns {config, env_steps, iteration}."""
    import json
    with np.load(path) as z:
        flat = {k: z[k] for k in z.files}
    extra = {"config": json.loads(bytes(flat.pop("config_json")).decode()) if "config_json" in flat else None,
             "env_steps": int(flat.pop("env_steps")) if "env_steps" in flat else None,
             "iteration": int(flat.pop("iteration")) if "iteration" in flat else None}
    tree = unflatten(flat)
    normalizer_from_flax(learner.normalizer, tree["normalizer"])
    policy_from_flax(learner.policy, tree["policy"])
    if "value" in tree:
        value_from_flax(learner.value, tree["value"])
    if load_optimizer and "optimizer" in tree:
        o = tree["optimizer"]
        with torch.no_grad():
            for name, buf in (("mu", learner.opt.exp_avg), ("nu", learner.opt.exp_avg_sq)):
                buf.zero_()                                   # (pad columns stay exactly zero)
                view = _moment_view(learner, buf)
                policy_from_flax(learner.policy, o[name]["policy"], view)
                value_from_flax(learner.value, o[name]["value"], view)
        learner.opt.t = int(o["count"])
    if load_optimizer and "rng" in tree:
        rng_from_tree(learner, tree["rng"])
    if hasattr(learner, "_refresh_padded_weights"):
        learner._refresh_padded_weights()
    return extra


# ---- the reference's checkpoint DIRECTORY layout (checkpointing.py:280-306: ocp.args.Composite(policy=StandardSave, train_state=StandardSave,
# config=JsonSave) under <directory>/<step>/) ------------------------------------------------------------------------------------------
def save_step_dir(directory, step: int, learner, config: dict | None = None, env_steps: int | None = None) -> str:
    """<directory>/<step>/{policy.npz, train_state.npz, config/metadata}: the three items of the reference's Composite save under the
    reference's names, `step` = the eval iteration (ppo.py:700-711 saves step 0, :787-795 step `it`).

    * `config/metadata` is the JSON file orbax's JsonCheckpointHandler writes for `JsonSave(config)`;
    * `policy` = the pair (normalizer_params, policy_params) the reference hands to StandardSave — here `policy.npz` with keys
      `0/{count,mean,summed_variance,std}` and `1/params/{encoder,decoder}/...` (a tuple's items are numbered by orbax);
    * `train_state` = brax-style TrainingState(optimizer_state, params{policy, value}, normalizer_params, env_steps) — `train_state.npz`.

    The two pytree items are NOT orbax containers: StandardSave writes a tensorstore OCDBT / zarr store whose byte format is defined by
    tensorstore (not in this image, not stated anywhere in the reference), so it cannot be restated or pinned here; INTEGRATION.md holds
    the five-line reference-side conversion (np.load -> unflatten -> ckpt_mgr.save).  The directory is built under a temporary name and
    renamed into place; an existing step is refused (as orbax does)."""
    import json
    import os
    import tempfile
    directory = str(directory)
    final = os.path.join(directory, str(int(step)))
    if os.path.exists(final):
        raise FileExistsError(f"checkpoint step {step} already exists in {directory}")
    os.makedirs(directory, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix=f".{int(step)}.tmp-", dir=directory)
    try:
        tree = learner_tree(learner)
        _atomic_savez(os.path.join(tmp, "policy.npz"), flatten({"0": tree["normalizer"], "1": tree["policy"]}), True)
        ts = {"optimizer_state": tree["optimizer"], "params": {"policy": tree["policy"], "value": tree["value"]},
              "normalizer_params": tree["normalizer"], "env_steps": np.asarray(0 if env_steps is None else env_steps, dtype=np.int64),
              "rng": tree["rng"], "iteration": np.asarray(int(step), dtype=np.int64)}
        _atomic_savez(os.path.join(tmp, "train_state.npz"), flatten(ts), True)
        os.makedirs(os.path.join(tmp, "config"))
        with open(os.path.join(tmp, "config", "metadata"), "w") as f:
            json.dump(config if config is not None else {}, f, default=str)
        os.rename(tmp, final)
    except BaseException:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        raise
    return final


def latest_step(directory) -> int | None:
    """orbax CheckpointManager.latest_step(): the largest all-digit sub-directory name."""
    import os
    try:
        steps = [int(d) for d in os.listdir(str(directory)) if d.isdigit() and os.path.isdir(os.path.join(str(directory), d))]
    except FileNotFoundError:
        return None
    return max(steps) if steps else None


def load_step_dir(path, learner, load_optimizer: bool = True) -> dict:
    """Restore from <directory>/<step>/ (or from <directory>: its latest step) written by save_step_dir; returns {config, env_steps, iteration}
    (checkpointing.load_training_state restores the whole TrainingState incl. env_steps: ppo.py:561-567)."""
    import json
    import os
    path = str(path)
    if not os.path.exists(os.path.join(path, "train_state.npz")):
        st = latest_step(path)
        if st is None:
            raise FileNotFoundError(f"no checkpoint step under {path}")
        path = os.path.join(path, str(st))
    with np.load(os.path.join(path, "train_state.npz")) as z:
        ts = unflatten({k: z[k] for k in z.files})
    normalizer_from_flax(learner.normalizer, ts["normalizer_params"])
    policy_from_flax(learner.policy, ts["params"]["policy"])
    value_from_flax(learner.value, ts["params"]["value"])
    if load_optimizer:
        o = ts["optimizer_state"]
        with torch.no_grad():
            for name, buf in (("mu", learner.opt.exp_avg), ("nu", learner.opt.exp_avg_sq)):
                buf.zero_()
                view = _moment_view(learner, buf)
                policy_from_flax(learner.policy, o[name]["policy"], view)
                value_from_flax(learner.value, o[name]["value"], view)
        learner.opt.t = int(o["count"])
        if "rng" in ts:
            rng_from_tree(learner, ts["rng"])
    if hasattr(learner, "_refresh_padded_weights"):
        learner._refresh_padded_weights()
    cfg = None
    if os.path.exists(os.path.join(path, "config", "metadata")):
        with open(os.path.join(path, "config", "metadata")) as f:
            cfg = json.load(f)
    return {"config": cfg, "env_steps": int(ts["env_steps"]), "iteration": int(ts["iteration"])}


def restore(path, learner, load_optimizer: bool = True) -> dict:
    """A .npz file, a step directory, or a checkpoint directory (latest step)."""
    import os
    return load_step_dir(path, learner, load_optimizer) if os.path.isdir(str(path)) else load_npz(path, learner, load_optimizer)
