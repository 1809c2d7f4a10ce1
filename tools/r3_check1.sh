#!/bin/bash
# round-3 first GPU call: new GPU tests, the self-launch path with one RCCL rank, a baseline bench line of the unchanged kernels
set -u
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py -m gpu -x -q -k "twice or retained or resume or train_entry or checkpoint" > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r3a/tests.log
TMJX_FORCE_SPAWN=1 TMJX_COLLECTIVES_ALWAYS=1 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_selflaunch.json 2> gpurun_out/r3a/bench_selflaunch.err; echo "selflaunch rc=$?"
tail -3 gpurun_out/r3a/bench_selflaunch.err
python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r3a/bench_selflaunch.json","gpurun_out/r3a/bench.json"):
    try:
        o=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, o["value"], o["ms_per_step"], o["config"]["ranks_seen"], o["config"]["rollout_ms_per_step"], o["config"]["sgd_ms_per_step"], o["config"]["rollout_only_env_steps_per_s_per_gpu"], o.get("mjx_cpu"), o.get("so_build_id"))
    except Exception as e: print(f, "ERR", e)
PY
