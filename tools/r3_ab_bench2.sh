#!/bin/bash
# tools/r3_ab_bench2.sh <tag> <old.so> <new.so>: the LDS-free linear / acting-path GPU tests with the new build, then alternating bench lines (REPS each)
TAG=$1; OLD=$2; NEW=$3
mkdir -p gpurun_out/$TAG
TMJX_SO=$NEW timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_free or inference or pipelined or rollout_store" > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/$TAG/tests.log
for rep in $(seq 1 ${REPS:-3}); do
for so in $OLD $NEW; do
  TMJX_SO=$so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$so rep$rep value %.0f  rollout_ms %.2f sgd_ms %.2f  rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))" | tee -a gpurun_out/$TAG/ab.txt
done
done
