#!/bin/bash
# A/B of an environment toggle on the SGD half of the bench configurations: bash tools/sgd_ab.sh <VAR> "<cfgs>"  (VAR=0 against VAR=1)
VAR=$1; CFGS=${2:-"cfg5 cfg4 cfg2"}
mkdir -p gpurun_out/sgd_ab
for cfg in $CFGS; do for v in 0 1 0 1; do
  env $VAR=$v python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs 2> gpurun_out/sgd_ab/err_${cfg}_$v.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('$cfg $VAR=$v', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"
done; done
