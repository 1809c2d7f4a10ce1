#!/bin/bash
# roll-out pipelining depth: the bench line with 2, 3 (n/a unless divisible), 4 env groups per GPU, same box
mkdir -p gpurun_out/groups
for g in 2 4 2 4 8; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --pipeline $g 2>gpurun_out/groups/g$g.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('groups $g value %.0f  rollout_ms %.1f sgd_ms %.1f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step']))" >> gpurun_out/groups/summary.txt
done
cat gpurun_out/groups/summary.txt
