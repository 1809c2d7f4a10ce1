#!/bin/bash
# tools/r3_ab_bench.sh <tag> <new.so> <old.so>: GPU suite with the new build, then the bench line of both builds alternately (same box)
TAG=$1; NEW=$2; OLD=$3
mkdir -p gpurun_out/$TAG
TMJX_SO=$NEW timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/$TAG/tests.log
for rep in 1 2; do
for so in $OLD $NEW; do
  TMJX_SO=$so python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$so value %.0f rollout_ms %.1f sgd_ms %.2f rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))"
done
done
