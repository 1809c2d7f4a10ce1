#!/bin/bash
# round 5, call 11: do pending workgroups of a physics dispatch (more envs than wave slots) hold up the other queues' small kernels?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
for n in 4096 3072 3584 2048; do
  python bench.py --envs-per-gpu $n --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('envs $n: value %.0f  rollout_ms %.1f (%.1f us per control step and env-thousand) sgd_ms %.1f  rollout_only %.0f k2_launch_ms %.3f' % (d['value'], c['rollout_ms_per_step'], c['rollout_ms_per_step']*1e3/80/($n/1000), c['sgd_ms_per_step'], c['rollout_only_env_steps_per_s_per_gpu'] or 0, d['roofline']['avg_launch_ms']))"
done | tee $O/envcount.txt
rocprofv3 --kernel-trace --output-format rocpd -d $O/trace -o x -- python3 bench.py --envs-per-gpu 3072 --steps 2 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $O/bench_3072.json 2> $O/err.txt
DB=$(find $O/trace -name "*.db" | head -1)
python3 tools/rollout_timeline.py $DB 200 > $O/rollout_timeline_3072.txt 2>&1; cut -c1-120 $O/rollout_timeline_3072.txt | tail -26
rm -rf $O/trace
