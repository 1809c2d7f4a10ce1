#!/bin/bash
# round 5, call u: the bf16 weight gradients as one grouped launch (tmjx_bgemm_dw_grouped): tests, then config 5 A/B (TMJX_BDW_GROUPED=0/1) and group budgets
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_gemm_bf16.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5u_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5u_tests.txt
run() { env "$@" python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs --no-live-pmc 2> gpurun_out/r5u_err.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('cfg5 $*', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"; }
for rep in 1 2; do run TMJX_BDW_GROUPED=0; run TMJX_BDW_GROUPED=1; done
for w in 768 1024 2048 3072; do run TMJX_BDW_GROUPED=1 TMJX_BDW_GROUP_WGS=$w; done
