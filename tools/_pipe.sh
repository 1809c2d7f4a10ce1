for cfg in "2 X=1" "2 GPU_MAX_HW_QUEUES=8" "3 GPU_MAX_HW_QUEUES=8" "4 GPU_MAX_HW_QUEUES=8" "4 GPU_MAX_HW_QUEUES=16"; do
  set -- $cfg
  echo "== pipeline $1 $2"; env $2 python bench.py --pipeline $1 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.0f rollout_ms %.1f sgd_ms %.1f rollout_only %.0f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0))" || tail -5 gpurun_out/err.txt
done
