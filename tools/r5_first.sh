#!/bin/bash
# round 5, first GPU call: new tests, cfg3 baseline (before M-aware tiling), default bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity_strict.py tests/test_gpu_parity.py -m gpu -x -q -k "rccl or frame or policy_params_fn or golden or side_effect" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
timeout -k 10 200 python bench.py --config cfg3 --steps 10 --warmup 2 --no-cpu-baseline --no-rollout-only > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "cfg3 rc=$?"
python3 - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r5a/bench_cfg3.json") if l.startswith("{")][-1]); c=o["config"]
print("cfg3", round(o["value"]), o["ms_per_step"], "rollout", c["rollout_ms_per_step"], "sgd/mb", c["sgd_ms_per_minibatch_step"], c["ranks_seen"], c["collectives"], c["minibatch_gemm_rows"])
PY
for W in 256 128 64; do echo "TMJX_DW_WGS=$W"; TMJX_DW_WGS=$W timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1; done
timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1
timeout -k 10 200 python tools/gemm_bench.py cfg2 5120 > $O/gemm_5120.txt 2>&1; cat $O/gemm_5120.txt
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; cut -c1-600 $O/bench_default.json
