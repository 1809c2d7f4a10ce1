#!/bin/bash
# round 5, call q: the physics unit rebuilt with every v_cndmask_b32_e32 (implicit vcc) re-encoded as _e64 (tools/micro/issue_rate3.hip: back-to-back
# e32 forms cost 13 - 19 cycles each on gfx950, the e64 encoding 3 - 5): parity of the one build against the other, then A/B timing
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5q; mkdir -p $O
A=track_mjx_amd/libtmjx_hip.so; B=${1:-alt/libtmjx_e64.so}
TMJX_SO=$B timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_strict.py -m gpu -x -q > $O/parity_b.txt 2>&1; echo "parity with $B rc=$?"; tail -3 $O/parity_b.txt
for rep in 1 2 3; do
  for v in $A $B; do
    echo "$v rep$rep $(TMJX_SO=$v python tools/time_step.py --steps 40 --scale 0.3 2>&1 | grep block)"
  done
done
for v in $A $B $A $B; do
  TMJX_SO=$v python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-live-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))"
done
