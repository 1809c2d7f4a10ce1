"""`python -m track_mjx_amd.train key=value ...` — mirror of the reference entrypoint
(`python -m track_mjx.train`, track_mjx/train.py:56-363): builds walker, reward config, clips, env and the
network sizes from the config and calls ppo.train.  hydra/wandb/orbax/rendering are out of scope;
the derived quantities follow train.py:221-225 (episode_length) and :298-301 (num_evals, num_resets_per_eval).

Multi-GPU: one process per GPU.  `python -m track_mjx_amd.train num_gpus=N ...` starts its own N ranks as a child
`python -m torch.distributed.run` (launch.py; the reference gets all local devices from one process via jax.pmap, ppo.py:409);
started under a launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.
`train_config.num_envs` is the GLOBAL env count as in the reference and is sharded across ranks.
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist

from . import clips as _clips
from . import config as _config
from .agent import ppo
from .environment import MultiClipTracking, RewardConfig
from .walker import Rodent


def load_clip_sets(cfg: dict, n_synthetic_clips: int = 64, clip_seed: int = 0):
    """(train_clips, test_clips or None) as track_mjx/train.py:163-209 selects them: `data_path` = an HDF5 clip file in the
    stac-mjx or ReferenceClip layout (io/load.py) or "synthetic"; a JSON split file (`train_setup.train_test_split_info`), a
    random split (`train_subset_ratio` with a file) or the whole set."""
    import json
    from .io import load
    ts = cfg["train_setup"]
    path = cfg.get("data_path", "synthetic")
    if path == "synthetic":
        walker = Rodent(**cfg["walker_config"])
        return _clips.make_synthetic_clips(walker.model, n_synthetic_clips, n_frames=cfg["reference_config"]["clip_length"], seed=clip_seed,
                                           mocap_hz=cfg["env_config"]["env_args"]["mocap_hz"]), None
    all_clips = load.load_data(path)
    if ts.get("train_test_split_info") is not None:
        with open(ts["train_test_split_info"], "r") as f:
            split = json.load(f)
        train_idx = split["train"] if ts.get("train_subset_ratio") is None else split["train_subset"][f"{ts['train_subset_ratio']:.2f}"]
        return load.select_clips(all_clips, train_idx), load.select_clips(all_clips, split["test"])
    if ts.get("train_subset_ratio") is not None:
        return load.generate_train_test_split(all_clips, test_ratio=1 - ts["train_subset_ratio"])
    return all_clips, None


def build_env(cfg: dict, num_envs_local: int, device, n_clips: int = 64, clip_seed: int = 0, reference_clip=None, share_clips_with=None):
    """`reference_clip`: a clip table built earlier (env groups of one rank share it); None = generate the synthetic table.
    `share_clips_with`: an env of this rank built from the same table: its resident device copy is used instead of a second upload."""
    walker = Rodent(**cfg["walker_config"])
    reward_config = RewardConfig(**cfg["env_config"]["reward_weights"])
    if reference_clip is None:
        reference_clip = load_clip_sets(cfg, n_clips, clip_seed)[0]
    env = MultiClipTracking(reference_clip, walker, reward_config, **cfg["env_config"]["env_args"], **cfg["reference_config"],
                            num_envs=num_envs_local, device=device, share_clips_with=share_clips_with)
    return env


def main(argv=None, runner=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    full_argv = list(argv)
    cfg_path = None
    if argv and argv[0].endswith((".yaml", ".yml")):
        cfg_path = argv.pop(0)
    cfg_name = None      # hydra's `--config-name <stem>` / `--config-name=<stem>` (train.py:56): one of config.NAMED_CONFIGS
    for i, a in enumerate(list(argv)):
        if a == "--config-name" and i + 1 < len(argv):
            cfg_name = argv[i + 1]; del argv[i:i + 2]; break
        if a.startswith("--config-name="):
            cfg_name = a.split("=", 1)[1]; del argv[i]; break
    cfg = _config.load_config(cfg_path, argv, name=cfg_name)
    from . import launch
    num_gpus = int(cfg.get("num_gpus", 1))
    if launch.needs_spawn(num_gpus):        # before anything touches the GPU: the ranks are a child process, never an exec
        return launch.spawn_ranks(num_gpus, ["-m", "track_mjx_amd.train"], full_argv, runner=runner)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TMJX_REHEARSE_ON_ONE_GPU=1 (tools/gpu_lab.sh rehearse): every rank on cuda:0 with gloo collectives on the device tensors — the multi-rank path on
    # a one-GPU box, where RCCL refuses two ranks on a device; a plumbing check, not a way to train
    rehearse = bool(os.environ.get("TMJX_REHEARSE_ON_ONE_GPU"))
    device = torch.device("cuda:0" if rehearse else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    tc = cfg["train_setup"]["train_config"]
    lo, hi = ppo.shard_range(int(tc["num_envs"]), int(os.environ.get("RANK", "0")), world)
    # `rollout_groups` env groups per rank (default: ppo.default_groups — 3 at 4096 envs; sizes: ppo.group_sizes): their roll-outs are pipelined on
    # separate HIP streams (agent/ppo.py: collect)
    nc_ = cfg["network_config"]
    widest = max(max(nc_[k]) for k in ("encoder_layer_sizes", "decoder_layer_sizes"))
    sizes = ppo.group_sizes(hi - lo, int(cfg.get("rollout_groups", 0)) or ppo.default_groups(hi - lo, device, widest_layer=widest))
    ngrp = len(sizes)
    train_clips, test_clips = load_clip_sets(cfg, int(cfg.get("n_synthetic_clips", 64)))
    envs = [build_env(cfg, sizes[0], device, reference_clip=train_clips)]
    envs += [build_env(cfg, sizes[k], device, reference_clip=train_clips, share_clips_with=envs[0]) for k in range(1, ngrp)]      # one clip upload per rank
    env = envs[0]
    rc, ts = cfg["reference_config"], cfg["train_setup"]
    # train.py:221-225
    episode_length = (rc["clip_length"] - rc["random_init_range"] - rc["traj_length"]) * env._steps_for_cur_frame
    nc = cfg["network_config"]
    num_evals = int(tc["num_timesteps"] / ts["eval_every"])
    num_resets_per_eval = ts["eval_every"] // ts["reset_every"]

    # evaluator env (ppo.py:629-647: the same environment class with num_eval_envs envs; rank 0 only)
    num_eval_envs = int(tc.get("num_eval_envs", 128))
    # evaluated on the held-out clips when there is a split, else on the training set (train.py:176-209)
    eval_env = build_env(cfg, num_eval_envs, device, reference_clip=test_clips if test_clips is not None and test_clips.position.shape[0] > 0 else train_clips) \
        if int(os.environ.get("RANK", "0")) == 0 and num_eval_envs > 0 else None

    def progress(num_steps, metrics):
        print(f"[train] steps={num_steps} " + " ".join(f"{k}={v:.4g}" for k, v in sorted(metrics.items())), flush=True)

    ppo.train(envs if ngrp > 1 else env, num_timesteps=tc["num_timesteps"], episode_length=int(episode_length), num_evals=num_evals,
              num_resets_per_eval=num_resets_per_eval, learning_rate=tc["learning_rate"], entropy_cost=tc["entropy_cost"],
              discounting=tc["discounting"], seed=tc["seed"], unroll_length=tc["unroll_length"], batch_size=tc["batch_size"],
              num_minibatches=tc["num_minibatches"], num_updates_per_batch=tc["num_updates_per_batch"],
              normalize_observations=tc["normalize_observations"], reward_scaling=tc["reward_scaling"],
              clipping_epsilon=tc["clipping_epsilon"], kl_weight=nc["kl_weight"], use_kl_schedule=nc["kl_schedule"],
              encoder_hidden_layer_sizes=nc["encoder_layer_sizes"], decoder_hidden_layer_sizes=nc["decoder_layer_sizes"],
              value_hidden_layer_sizes=nc["critic_layer_sizes"], intention_latent_size=nc["intention_size"], progress_fn=progress,
              max_training_steps=cfg.get("max_training_steps"), eval_env=eval_env, num_eval_envs=num_eval_envs,
              deterministic_eval=bool(tc.get("deterministic_eval", False)), config_dict=cfg, action_repeat=int(tc.get("action_repeat", 1)),
              checkpoint_path=cfg.get("checkpoint_path"), restore_from=cfg.get("restore_from"), shuffle_rng=str(cfg.get("shuffle_rng", "torch")), act_rng=str(cfg.get("act_rng", "device")),
              matmul_dtype=torch.bfloat16 if str(cfg.get("mlp_gemm_inputs", "f32")).lower() in ("bf16", "bfloat16") else None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
