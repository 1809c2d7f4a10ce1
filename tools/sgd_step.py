#!/usr/bin/env python3
"""The SGD half of a PPO training step alone (64 minibatch updates on a filled roll-out buffer), eager launches (no hipGraph):
the run rocprofv3 collects the MFMA counters of the MLP GEMMs over.  `--config` as in bench.py; `--envs` shrinks the roll-out."""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench as _bench  # noqa: E402
from track_mjx_amd import config as _config  # noqa: E402
from track_mjx_amd.agent import ppo  # noqa: E402
from track_mjx_amd.environment import wrap  # noqa: E402
from track_mjx_amd.train import build_env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=sorted(_bench.CONFIGS))
    ap.add_argument("--updates", type=int, default=1, help="num_updates_per_batch passes (16 minibatch steps each)")
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    bc = _bench.CONFIGS[args.config]
    dev = torch.device("cuda:0")
    cfg = _config.default_config()
    cfg["network_config"].update(**bc["nets"])
    tc, nc = cfg["train_setup"]["train_config"], cfg["network_config"]
    n = bc["envs_per_gpu"]
    env = wrap(build_env(cfg, n, dev, n_clips=min(bc["n_clips"], 64)), episode_length=195)
    L = ppo.PPOLearner(env, encoder_layers=nc["encoder_layer_sizes"], decoder_layers=nc["decoder_layer_sizes"], critic_layers=nc["critic_layer_sizes"],
                       latents=nc["intention_size"], unroll_length=tc["unroll_length"], batch_size=bc["rows_per_gpu"],
                       num_minibatches=tc["num_minibatches"], num_updates_per_batch=args.updates, kl_weight=nc["kl_weight"], seed=0,
                       matmul_dtype=torch.bfloat16 if bc["matmul_dtype"] == "bf16" else None, use_graph=args.graph)
    L.states[0] = env.reset(torch.Generator().manual_seed(0))
    # a synthetic roll-out buffer (the counters are about the GEMMs, not the data): unit-normal observations / actions
    g = torch.Generator(device=dev).manual_seed(0)
    for k, v in L.buf.items():
        v.copy_(torch.randn(v.shape, generator=g, device=dev) if k not in ("discount", "truncation") else torch.ones_like(v) * (k == "discount"))
    L.buf["log_prob"].fill_(-30.0)
    L.update(0)
    torch.cuda.synchronize()
    t0 = time.time()
    L.update(0)
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps = args.updates * tc["num_minibatches"]
    P = sum(p.numel() for p in L.policy.parameters() if p.dim() == 2)
    V = sum(p.numel() for p in L.value.parameters() if p.dim() == 2)
    rows = L.local_batch * L.T
    flops = 2.0 * 3.0 * (P + V) * rows * steps
    print(f"config={args.config} minibatch_steps={steps} rows/step={rows} ms/step={dt / steps * 1e3:.3f} GEMM TFLOP/s(fwd+bwd, 6 x weights x rows)={flops / dt / 1e12:.1f}", flush=True)


if __name__ == "__main__":
    main()
