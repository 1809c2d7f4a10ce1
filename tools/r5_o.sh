#!/bin/bash
# round 5, call 16: the fixed acting-layer kernel alone first, then the whole GPU suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5o; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds_tiled_acting" > $O/t1.txt 2>&1; rc=$?; echo "acting-layer test rc=$rc"; tail -2 $O/t1.txt
[ $rc -eq 0 ] && { timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt; }
