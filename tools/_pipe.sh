for cfg in "2 TMJX_PRIO_STREAMS=1" "4 TMJX_PRIO_STREAMS=1" "3 TMJX_PRIO_STREAMS=1"; do
  set -- $cfg
  echo "== pipeline $1 $2"; env $2 python bench.py --pipeline $1 --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %.0f rollout_ms %.1f sgd_ms %.1f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step']))" || tail -5 gpurun_out/err.txt
done
