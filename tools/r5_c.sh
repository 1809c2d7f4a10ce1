#!/bin/bash
# round 5, call 3: group-level dW split — tests, cfg2 / cfg3 SGD step sweeps
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity_strict.py -m gpu -x -q -k "gemm or frame or dense or layernorm or value_net or weight" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for G in 512 256 1024 0; do
  echo "TMJX_DW_GROUP_WGS=$G"
  TMJX_DW_GROUP_WGS=$G timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1
  TMJX_DW_GROUP_WGS=$G timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1
done
TMJX_DW_GROUP_WGS=512 timeout -k 10 120 python tools/sgd_step.py --config cfg4 --graph --updates 2 2>&1 | tail -1
TMJX_DW_GROUP_WGS=0 timeout -k 10 120 python tools/sgd_step.py --config cfg4 --graph --updates 2 2>&1 | tail -1
timeout -k 10 200 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-rollout-only > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python3 - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r5c/bench_cfg3.json") if l.startswith("{")][-1]); c=o["config"]
print("cfg3", round(o["value"]), o["ms_per_step"], "rollout", c["rollout_ms_per_step"], "sgd/mb", c["sgd_ms_per_minibatch_step"], c["ranks_seen"], c["collectives"], c["minibatch_gemm_rows"])
PY
