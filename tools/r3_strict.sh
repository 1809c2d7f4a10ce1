#!/bin/bash
set -u
mkdir -p gpurun_out/r3d
timeout -k 10 900 python -m pytest tests/test_gpu_parity_strict.py -m gpu -q -s -k "worst_env" > gpurun_out/r3d/strict.log 2>&1; echo "rc=$?"
grep -A8 "^\[scale" gpurun_out/r3d/strict.log | cut -c1-400; tail -3 gpurun_out/r3d/strict.log
