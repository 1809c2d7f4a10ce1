#!/bin/bash
# round 5, call 4: K2 at 14 envs per CU (9 LDS granules, 128 VGPRs): parity tests, step time, bench, group splits
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_strict.py -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
python tools/time_step.py --envs 4096 --steps 20 --scale 0.3 2>&1 | tail -1
python tools/time_step.py --envs 3584 --steps 20 --scale 0.3 2>&1 | tail -1
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r5d/bench.json") if l.startswith("{")][-1]); c=o["config"]
print("cfg2", round(o["value"]), o["ms_per_step"], "rollout", c["rollout_ms_per_step"], "sgd/mb", c["sgd_ms_per_minibatch_step"], "rollout-only", c["rollout_only_env_steps_per_s_per_gpu"], "k2 ms", o["roofline"]["avg_launch_ms"], o["roofline"]["avg_launch_ms_isolated"])
PY
bash tools/group_sizes_ab.sh r5d "1368,1364,1364" "1792,1792,512" "1200,1200,1696" "2048,2048" "1024,1024,1024,1024" > $O/groups.txt 2>&1; cat $O/groups.txt
