#!/bin/bash
# cfg5 / cfg4 bench lines, strict parity log, one-rank self-launch, with the final build
set -u
mkdir -p gpurun_out/fin2
timeout -k 10 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/fin2/bench_cfg5.json 2> gpurun_out/fin2/bench_cfg5.err; echo "cfg5 rc=$?"
timeout -k 10 600 python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/fin2/bench_cfg4.json 2> gpurun_out/fin2/bench_cfg4.err; echo "cfg4 rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_parity_strict.py -m gpu -q -s > gpurun_out/fin2/strict.log 2>&1; echo "strict rc=$?"
TMJX_FORCE_SPAWN=1 TMJX_COLLECTIVES_ALWAYS=1 timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/fin2/bench_selflaunch.json 2> gpurun_out/fin2/bench_selflaunch.err; echo "selflaunch rc=$?"
python - <<'PY'
import json
for f in ("bench_cfg5","bench_cfg4","bench_selflaunch"):
    try:
        o=json.loads([l for l in open(f"gpurun_out/fin2/{f}.json") if l.startswith("{")][-1]); c=o["config"]
        print(f, round(o["value"]), round(o["ms_per_step"],1), "rollout", round(c["rollout_ms_per_step"],1), "sgd", round(c["sgd_ms_per_step"],1), "mb", round(c["sgd_ms_per_minibatch_step"],3), o["dtype"], c.get("ranks_seen"))
    except Exception as e: print(f, "ERR", e)
PY
grep -E "passed|failed" gpurun_out/fin2/strict.log
