#!/bin/bash
# round 5, call 10: roll-out timeline with priorities; direct [row][n_env] access instead of the env-major record
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5j; mkdir -p $O
for rep in 1 2; do
for mode in rec norec; do
  if [ $mode = norec ]; then export TMJX_NO_RECORD=1; else unset TMJX_NO_RECORD; fi
  python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$mode rep$rep: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f k2_launch_ms %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, d['roofline']['avg_launch_ms']))"
done; done | tee $O/record_ab.txt
unset TMJX_NO_RECORD
bash tools/gpu_lab.sh timeline cfg2 > $O/timeline.log 2>&1; cp gpurun_out/timeline/cfg2_rollout_timeline.txt gpurun_out/timeline/cfg2_sgd_step_timeline.txt $O/ 2>/dev/null; cat $O/cfg2_rollout_timeline.txt | cut -c1-120
