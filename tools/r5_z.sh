#!/bin/bash
# round 5, call z: 1-wide layers (the value MLP's head) as matrix-vector kernels in every pass (tmjx_head_fwd / tmjx_head_dw): tests, then SGD A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_gemm_bf16.py tests/test_gpu_parity.py tests/test_gpu_rccl.py -m gpu -x -q > gpurun_out/r5z2_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5z2_tests.txt
for cfg in cfg2 cfg3 cfg5; do for rep in 1 2; do for v in 0 1; do
  env TMJX_HEAD_KERNELS=$v python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs --no-live-pmc 2> gpurun_out/r5z_err.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('$cfg TMJX_HEAD_KERNELS=$v', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"
done; done; done
