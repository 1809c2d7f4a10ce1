for c in cfg4 cfg5; do
python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only 2>gpurun_out/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$c value %.0f rollout_ms %.1f sgd_ms %.1f mfma %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_step'], d['roofline_mfma']['frac']))" || tail -5 gpurun_out/err.txt
done
