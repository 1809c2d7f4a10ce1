#!/bin/bash
# two and four ranks of bench.py on ONE GPU (gloo collectives on device tensors): the N > 1 path of the bench end to end.  GPU box.
set -u
mkdir -p gpurun_out/rehearse
for N in 2 4; do
  TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
    bench.py --gpus $N --envs-per-gpu 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-rollout-only > gpurun_out/rehearse/n$N.json 2> gpurun_out/rehearse/n$N.err
  echo "N=$N rc=$?"
  python - gpurun_out/rehearse/n$N.json <<'PY'
import json, sys
ls=[l for l in open(sys.argv[1]) if l.startswith("{")]
print(len(ls), "JSON line(s)")
if ls:
    o=json.loads(ls[-1]); c=o["config"]
    print(o["n_gpus"], round(o["value"]), c["ranks_seen"], c["parallelism"], c["global_batch"], c["env_steps_per_step"], c.get("rehearsal","")[:40])
PY
  tail -3 gpurun_out/rehearse/n$N.err
done
for N in 2 3; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29530 + N)) tools/two_rank_sync_check.py 3 > gpurun_out/rehearse/sync$N.txt 2> gpurun_out/rehearse/sync$N.err
  echo "sync N=$N rc=$?"; grep "ranks on one GPU" gpurun_out/rehearse/sync$N.txt; grep -i "error\|assert" gpurun_out/rehearse/sync$N.err | head -5
done
# python -m track_mjx_amd.train with two ranks (num_envs is global: 256 -> 128 per rank), evaluator on rank 0, checkpoint written by rank 0
rm -rf /tmp/tmjx_rehearse_ckpt
TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 -m track_mjx_amd.train \
  train_setup.train_config.num_envs=256 train_setup.train_config.batch_size=64 train_setup.train_config.num_minibatches=4 train_setup.train_config.unroll_length=5 \
  train_setup.train_config.num_updates_per_batch=2 "network_config.encoder_layer_sizes=[64,64]" "network_config.decoder_layer_sizes=[64,64]" "network_config.critic_layer_sizes=[64,64]" \
  train_setup.train_config.num_timesteps=100000 train_setup.eval_every=50000 train_setup.reset_every=50000 max_training_steps=3 n_synthetic_clips=4 num_gpus=2 \
  checkpoint_path=/tmp/tmjx_rehearse_ckpt > gpurun_out/rehearse/train2.txt 2> gpurun_out/rehearse/train2.err
echo "train 2 ranks rc=$?"; grep "^\[train\]" gpurun_out/rehearse/train2.txt | cut -c1-260 | tail -3; ls /tmp/tmjx_rehearse_ckpt 2>/dev/null | head; grep -i "error\|assert" gpurun_out/rehearse/train2.err | head -5
