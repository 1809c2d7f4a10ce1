#!/bin/bash
# round 5, call 9: wave priority of the serial-phase kernels (s_setprio 3) against the same build without it: bench line A/B, alternating
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5i; mkdir -p $O
for rep in 1 2 3; do
for v in track_mjx_amd/libtmjx_hip.so alt/libtmjx_noprio.so; do
  TMJX_SO=$v python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v rep$rep: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f k2_launch_ms %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, d['roofline']['avg_launch_ms']))"
done; done | tee $O/prio_ab.txt
TMJX_SO=track_mjx_amd/libtmjx_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or autoreset or pipelined or inference" 2>&1 | tail -2
