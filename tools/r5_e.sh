#!/bin/bash
# round 5, call 5: bisect the action-repeat mismatch and A/B the kernel variants (base = before the LDS moves, m1 = packed dof words only,
# lb3 = all moves at 3 waves per SIMD, cur = all moves at 4 waves per SIMD / 14 envs per CU)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
for v in alt/libtmjx_base.so alt/libtmjx_m1.so alt/libtmjx_lb3.so track_mjx_amd/libtmjx_hip.so; do
  for rep in 1 2; do
  TMJX_SO=$v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "action_repeat" > $O/t_$(basename $v).txt 2>&1; echo "$v action_repeat rc=$?"
  done
done
TMJX_SO=track_mjx_amd/libtmjx_hip.so timeout -k 10 300 python tools/determinism_check.py > $O/determinism.txt 2>&1; tail -5 $O/determinism.txt
REPS=3 bash tools/ab_k2.sh r5e alt/libtmjx_base.so alt/libtmjx_m1.so alt/libtmjx_lb3.so track_mjx_amd/libtmjx_hip.so
for v in alt/libtmjx_base.so alt/libtmjx_lb3.so track_mjx_amd/libtmjx_hip.so; do
  echo "$v exact fill 12/CU (3072 envs): $(TMJX_SO=$v python tools/time_step.py --envs 3072 --steps 40 --scale 0.3 2>&1 | grep block)"
  echo "$v 6144 envs: $(TMJX_SO=$v python tools/time_step.py --envs 6144 --steps 30 --scale 0.3 2>&1 | grep block)"
done
echo "cur exact fill 14/CU (3584 envs): $(python tools/time_step.py --envs 3584 --steps 40 --scale 0.3 2>&1 | grep block)"
echo "cur padded to 12/CU, 3072 envs: $(TMJX_LDS_PAD_KB=1 python tools/time_step.py --envs 3072 --steps 40 --scale 0.3 2>&1 | grep block)"
for v in alt/libtmjx_base.so alt/libtmjx_lb3.so track_mjx_amd/libtmjx_hip.so; do
  TMJX_SO=$v python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v: value %.0f  rollout_ms %.1f sgd_ms %.1f  rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))"
done
