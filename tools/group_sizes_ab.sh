#!/bin/bash
# bench line (training + roll-out-only) for several env-group splits of the 4096 envs: bash tools/group_sizes_ab.sh <tag> "1368,1364,1364" "1536,1536,1024" ...
TAG=$1; shift
mkdir -p gpurun_out
for gs in "$@"; do
  TMJX_GROUP_SIZES=$gs python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('groups $gs: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))" >> gpurun_out/groups_$TAG.txt
done
cat gpurun_out/groups_$TAG.txt
