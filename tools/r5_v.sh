#!/bin/bash
# round 5, call v: grouped bf16 weight gradients with slabs in eights: tests + config 5 A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_gemm_bf16.py -m gpu -x -q > gpurun_out/r5v_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5v_tests.txt
run() { env "$@" python bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs --no-live-pmc 2> gpurun_out/r5v_err.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('cfg5 $*', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"; }
for rep in 1 2 3; do run TMJX_BDW_GROUPED=0; run TMJX_BDW_GROUPED=1; done
run TMJX_BDW_GROUPED=1 TMJX_BDW_GROUP_WGS=4096
