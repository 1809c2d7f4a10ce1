#!/bin/bash
# A/B of environment switches on the GPU box: tools/ab_env.sh "<ENV=1 ...>" "<ENV=...>" ...  — one bench line (value, roll-out ms, SGD ms) per setting
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== [$cfg]"; env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.0f rollout_ms %.1f sgd_ms %.1f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step']))" || tail -5 gpurun_out/err.txt
done
