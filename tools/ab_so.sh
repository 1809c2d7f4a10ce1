#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_so.sh <tag> libA.so libB.so ...  — env.step timing (one launch of 4096 envs) at two action
# scales and the bench line (training + roll-out-only rates) per build, same box, same process order
TAG=$1; shift
mkdir -p gpurun_out
for so in "$@"; do
  echo "== $so" >> gpurun_out/ab_$TAG.txt
  TMJX_SO=$so python tools/time_step.py --steps 40 --scale 0.3 2>&1 | grep block >> gpurun_out/ab_$TAG.txt
  TMJX_SO=$so python tools/time_step.py --steps 40 2>&1 | grep block >> gpurun_out/ab_$TAG.txt
  TMJX_SO=$so python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('bench value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f  k2_shared_ms %.3f k2_isolated_ms %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, r['avg_launch_ms'], r['avg_launch_ms_isolated'] or 0))" >> gpurun_out/ab_$TAG.txt
done
cat gpurun_out/ab_$TAG.txt
