#!/bin/bash
# round 5, call 2: M-aware GEMM tiles — tests, GEMM micro-bench at 5 120 rows, cfg3 SGD step and its timeline
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity_strict.py -m gpu -x -q -k "gemm or frame or dense or layernorm or value_net or weight" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout -k 10 200 python tools/gemm_bench.py cfg2 5120 > $O/gemm_5120.txt 2>&1; tail -3 $O/gemm_5120.txt
TMJX_GEMM_MT=5 timeout -k 10 200 python tools/gemm_bench.py cfg2 5120 > $O/gemm_5120_mt5.txt 2>&1; tail -3 $O/gemm_5120_mt5.txt
for W in 256 96 48; do echo "TMJX_DW_WGS=$W"; TMJX_DW_WGS=$W timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1; done
echo "MT=5:"; TMJX_GEMM_MT=5 timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1
timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1
bash tools/gpu_lab.sh timeline cfg3 > $O/timeline.txt 2>&1; cp gpurun_out/timeline/cfg3_sgd_step_timeline.txt $O/ 2>/dev/null; cat $O/cfg3_sgd_step_timeline.txt | cut -c1-150
