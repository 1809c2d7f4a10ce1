#!/bin/bash
# round 5, call 14: the acting policy's layers through the 20 KB LDS tile (tmjx_linear_act) and the register-capped LDS-free kernel: tests, bench A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm_bf16.py -m gpu -x -q -k "acting or lds_free or inference or rollout or pipelined or nolds or act" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for rep in 1 2 3; do
for mode in 1 0; do
  TMJX_ACT_LDS=$mode python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('TMJX_ACT_LDS=$mode rep$rep: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f k2_launch_ms %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, d['roofline']['avg_launch_ms']))"
done; done | tee $O/act_lds_ab.txt
TMJX_SO=alt/libtmjx_noprio.so python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('previous build (no priority, 140-register LDS-free layers): value %.0f  rollout_ms %.1f sgd_ms %.1f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step']))" | tee -a $O/act_lds_ab.txt
for cfg in cfg4 cfg3; do for mode in 1 0; do TMJX_ACT_LDS=$mode python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-rollout-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$cfg TMJX_ACT_LDS=$mode: value %.0f  rollout_ms %.1f sgd_ms %.1f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step']))"; done; done | tee -a $O/act_lds_ab.txt
bash tools/gpu_lab.sh timeline cfg2 > $O/timeline.log 2>&1; cut -c1-120 gpurun_out/timeline/cfg2_rollout_timeline.txt | tail -26
