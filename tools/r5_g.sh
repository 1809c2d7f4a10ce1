#!/bin/bash
# round 5, call 7: full GPU suite on the current build + bench line + dW group target sweep for cfg2
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
for G in 1024 1536 2048 768 1024; do echo "TMJX_DW_GROUP_WGS=$G $(TMJX_DW_GROUP_WGS=$G timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1)"; done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r5g/bench.json") if l.startswith("{")][-1]); c=o["config"]
print("cfg2", round(o["value"]), o["ms_per_step"], "rollout", c["rollout_ms_per_step"], "sgd/mb", c["sgd_ms_per_minibatch_step"], "rollout-only", c["rollout_only_env_steps_per_s_per_gpu"], "k2 ms", o["roofline"]["avg_launch_ms"], o["roofline"]["avg_launch_ms_isolated"])
PY
