#!/bin/bash
# HIP hardware-queue count x env groups: bench lines (training + roll-out only) per setting, one box
mkdir -p gpurun_out/hwq; rm -f gpurun_out/hwq/ab.txt
run() { # label, env assignments..., -- bench args
  label=$1; shift
  env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$label value %.0f  rollout_ms %.2f sgd_ms %.2f  rollout_only %.0f groups %d' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, c['concurrent_physics_launches']))" | tee -a gpurun_out/hwq/ab.txt
}
run "default-q g3" TMJX_X=0
run "q8 g3" GPU_MAX_HW_QUEUES=8
run "q8 g4" GPU_MAX_HW_QUEUES=8 TMJX_GROUP_SIZES=1024,1024,1024,1024
run "q8 g6" GPU_MAX_HW_QUEUES=8 TMJX_GROUP_SIZES=684,684,684,684,680,680
run "q2 g3" GPU_MAX_HW_QUEUES=2
run "default-q g3" TMJX_X=0
