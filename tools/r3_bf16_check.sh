#!/bin/bash
set -u
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_gemm_bf16.py -m gpu -x -q > gpurun_out/r3b/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r3b/tests.log
