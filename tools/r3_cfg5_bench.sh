#!/bin/bash
# cfg5 / cfg4 bench lines + bf16 unit tests.  usage: bash tools/r3_cfg5_bench.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/cfg5b_$TAG
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_gemm_bf16.py tests/test_gpu_parity.py -m gpu -x -q -k "bgemm or bf16 or full_size or shadow" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout -k 10 300 python tools/sgd_step.py --config cfg5 --graph > $OUT/sgd_graph.log 2>&1; echo "rc=$?"; tail -1 $OUT/sgd_graph.log
timeout -k 10 600 python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "bench rc=$?"
python - "$OUT/bench_cfg5.json" <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = o["config"]
print("cfg5 value", round(o["value"]), "ms/step", round(o["ms_per_step"], 1), "rollout", round(c["rollout_ms_per_step"], 1), "sgd", round(c["sgd_ms_per_step"], 1), "per mb", round(c["sgd_ms_per_minibatch_step"], 3), "dtype", o["dtype"])
PY
