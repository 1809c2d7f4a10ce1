#!/bin/bash
# tools/r3_env_ab.sh <ENVVAR> [bench args]: the bench line with ENVVAR unset / set to 1, alternately (same box)
V=$1; shift
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export $V=1; else unset $V; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-rollout-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$V=$v $* value %.0f rollout_ms %.1f sgd_ms %.2f per-mb %.4f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['sgd_ms_per_minibatch_step']))"
done
done
