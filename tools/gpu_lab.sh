#!/bin/bash
# One parameterised script for the GPU-box measurements that are not an A/B of two builds (those: tools/gpu_ab.sh; K2 counters: tools/k2_profile.sh;
# env-group splits: tools/group_sizes_ab.sh).  Replaces round 3's per-experiment tools/r3_*.sh.  Everything lands under gpurun_out/<mode>/.
#   bash tools/gpu_lab.sh cfgstats            rocprofv3 kernel stats of the config-4 and config-5 bench commands (+ the count of library-GEMM rows)
#   bash tools/gpu_lab.sh timeline [cfgN]     rocprofv3 kernel trace (rocpd) of the bench -> one SGD minibatch step (tools/step_timeline.py) and,
#                                             for cfg2, one env group's serial roll-out phase (tools/rollout_timeline.py)
#   bash tools/gpu_lab.sh starve              the rehearsal's 4 ranks with every rank pinned to the same two host cores against unpinned: what host starvation does to the roll-out
#   bash tools/gpu_lab.sh rehearse            2 / 4 ranks of bench.py, tools/two_rank_sync_check.py and `python -m track_mjx_amd.train num_gpus=2`
#                                             on ONE GPU over gloo (TMJX_REHEARSE_ON_ONE_GPU=1): plumbing of the N > 1 path, never a measurement
set -u
MODE=${1:-}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$MODE; mkdir -p $OUT
case "$MODE" in
cfgstats)
  for CFG in cfg5 cfg4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -o x -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench_$CFG.json 2> $OUT/err_$CFG.txt
    echo "$CFG rc=$?"
    find $OUT/trace_$CFG -name "*kernel_stats.csv" -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
    rm -rf $OUT/trace_$CFG
    head -8 $OUT/${CFG}_kernel_stats.csv | cut -c1-140
    echo "Cijk rows: $(grep -c "Cijk" $OUT/${CFG}_kernel_stats.csv)"
  done;;
timeline)
  CFG=${1:-cfg2}
  rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -o x -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench_$CFG.json 2> $OUT/err_$CFG.txt
  DB=$(find $OUT/trace -name "*.db" | head -1)
  python3 tools/step_timeline.py $DB 10 > $OUT/${CFG}_sgd_step_timeline.txt 2>&1
  [ "$CFG" = cfg2 ] && python3 tools/rollout_timeline.py $DB 200 > $OUT/${CFG}_rollout_timeline.txt 2>&1 && cat $OUT/${CFG}_rollout_timeline.txt
  tail -40 $OUT/${CFG}_sgd_step_timeline.txt
  rm -rf $OUT/trace;;
rehearse)
  for N in 2 4; do
    TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
      bench.py --gpus $N --envs-per-gpu 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-rollout-only > $OUT/n$N.json 2> $OUT/n$N.err
    echo "N=$N rc=$?"; grep '^{' $OUT/n$N.json | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print(o['n_gpus'], round(o['value']), c['ranks_seen'], c['parallelism'], c['global_batch'], c.get('rehearsal','')[:40])"
  done
  for N in 2 3; do
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29530 + N)) tools/two_rank_sync_check.py 3 > $OUT/sync$N.txt 2> $OUT/sync$N.err
    echo "sync N=$N rc=$?"; grep "ranks on one GPU" $OUT/sync$N.txt
  done
  rm -rf /tmp/tmjx_rehearse_ckpt
  TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 -m track_mjx_amd.train \
    train_setup.train_config.num_envs=256 train_setup.train_config.batch_size=64 train_setup.train_config.num_minibatches=4 train_setup.train_config.unroll_length=5 \
    train_setup.train_config.num_updates_per_batch=2 "network_config.encoder_layer_sizes=[64,64]" "network_config.decoder_layer_sizes=[64,64]" "network_config.critic_layer_sizes=[64,64]" \
    train_setup.train_config.num_timesteps=100000 train_setup.eval_every=50000 train_setup.reset_every=50000 max_training_steps=3 n_synthetic_clips=4 num_gpus=2 \
    checkpoint_path=/tmp/tmjx_rehearse_ckpt > $OUT/train2.txt 2> $OUT/train2.err
  echo "train 2 ranks rc=$?"; grep "^\[train\]" $OUT/train2.txt | cut -c1-200 | tail -2;;
starve)
  # host-starvation stress of the roll-out: 4 ranks of bench.py sharing ONE GPU over gloo (the rehearsal mode: plumbing, never a measurement), once with the
  # ranks free to use every host core, once with all four pinned to the SAME two cores (TMJX_PIN_CORES: os.sched_setaffinity before the first GPU call) —
  # 4 ranks x 3 group streams x ~10 launches per group step from two cores.  What is compared: rollout_ms_per_step of the two runs
  for PIN in free 0,1; do
    [ "$PIN" = free ] && unset TMJX_PIN_CORES || export TMJX_PIN_CORES=$PIN
    TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29561 \
      bench.py --gpus 4 --envs-per-gpu 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-rollout-only > $OUT/pin_$PIN.json 2> $OUT/pin_$PIN.err
    echo "pinned to cores: $PIN  rc=$?"; grep '^{' $OUT/pin_$PIN.json | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('  ranks', c['ranks_seen'], ' roll-out ms per step', round(c['rollout_ms_per_step'], 2), ' sgd ms per step', round(c['sgd_ms_per_step'], 2), ' env-steps/s (all ranks on one GPU)', round(o['value']))"
  done;;
*) echo "usage: bash tools/gpu_lab.sh cfgstats | timeline [cfgN] | rehearse | starve"; exit 2;;
esac
