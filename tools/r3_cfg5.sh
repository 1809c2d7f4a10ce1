#!/bin/bash
# cfg5 (bf16 GEMM-input mode) SGD half: parity test, ms per minibatch step, rocprofv3 kernel stats.  usage: bash tools/r3_cfg5.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/cfg5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16 or full_size" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
timeout -k 10 300 python tools/sgd_step.py --config cfg5 --graph > $OUT/sgd_graph.log 2>&1; echo "rc=$?"; tail -2 $OUT/sgd_graph.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cfg5 -- python3 tools/sgd_step.py --config cfg5 > $OUT/sgd_eager.log 2>&1; echo "rc=$?"; tail -2 $OUT/sgd_eager.log
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -25 $OUT/kernel_stats.csv | cut -c1-200
