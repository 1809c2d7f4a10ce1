#!/bin/bash
# round 5, call 13: the value network's weight gradients as an early group on the side stream (TMJX_VALUE_DW_WGS budget; 0 = all in one group at the end)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5l; mkdir -p $O
for rep in 1 2; do
for V in 0 384 256 512 768; do echo "cfg2 TMJX_VALUE_DW_WGS=$V: $(TMJX_VALUE_DW_WGS=$V timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1)"; done
done
for V in 0 384; do echo "cfg3 TMJX_VALUE_DW_WGS=$V: $(TMJX_VALUE_DW_WGS=$V timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1)"; done
for V in 0 384; do echo "cfg4 TMJX_VALUE_DW_WGS=$V: $(TMJX_VALUE_DW_WGS=$V timeout -k 10 120 python tools/sgd_step.py --config cfg4 --graph --updates 2 2>&1 | tail -1)"; done
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_rccl.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm or grouped or rccl or loss_head or learner or train or deferred or weight" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
