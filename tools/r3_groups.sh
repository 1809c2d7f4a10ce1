#!/bin/bash
# env-group experiments (bench.py --pipeline / TMJX_GROUP_SIZES), same box.  Results of round 3 in DESIGN.md section 6.
mkdir -p gpurun_out/groups
rm -f gpurun_out/groups/summary.txt
run() {  # label, env assignments...
  L=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs $PIPE 2>gpurun_out/groups/$L.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$L value %.0f  rollout_ms %.1f sgd_ms %.1f rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))" >> gpurun_out/groups/summary.txt
}
PIPE="--pipeline 3" run g3 A=1
PIPE="--pipeline 3" run g3_1408_1408_1280 TMJX_GROUP_SIZES=1408,1408,1280
PIPE="--pipeline 2" run g2 A=1
PIPE="--pipeline 3" run g3 A=1
PIPE="--pipeline 3" run g3_1408_1408_1280 TMJX_GROUP_SIZES=1408,1408,1280
PIPE="--pipeline 3 --config cfg5 --no-rollout-only" run g3_cfg5 A=1
PIPE="--pipeline 2 --config cfg5 --no-rollout-only" run g2_cfg5 A=1
PIPE="--pipeline 3 --config cfg4 --no-rollout-only" run g3_cfg4 A=1
PIPE="--pipeline 2 --config cfg4 --no-rollout-only" run g2_cfg4 A=1
cat gpurun_out/groups/summary.txt
