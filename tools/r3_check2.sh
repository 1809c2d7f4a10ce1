#!/bin/bash
set -u
mkdir -p gpurun_out/r3c
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout or pipelined or train_ or evaluator or device_side or fused_inference" > gpurun_out/r3c/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r3c/tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r3c/bench.json") if l.startswith("{")][-1])
c=o["config"]
print(o["value"], o["ms_per_step"], "rollout", c["rollout_ms_per_step"], "sgd", c["sgd_ms_per_step"], "rollout-only", c["rollout_only_env_steps_per_s_per_gpu"], "k2 ms", o["roofline"]["avg_launch_ms"], o["roofline"]["avg_launch_ms_isolated"])
PY
TMJX_SO=track_mjx_amd/libtmjx_hip_prof.so python tools/phase_profile.py > gpurun_out/r3c/phase_profile.txt 2>&1; tail -36 gpurun_out/r3c/phase_profile.txt
