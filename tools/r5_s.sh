#!/bin/bash
# round 5, call s: the loss head in phases (B on the value network's stream, D off the main stream): tests, then SGD A/B on config 2 and config 3
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=${1:-TMJX_PPO_PHASES}; TESTS=${2:-"tests/test_gpu_parity.py tests/test_gpu_rccl.py"}
timeout -k 10 600 python -m pytest $TESTS -m gpu -x -q > gpurun_out/r5s_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5s_tests.txt
for cfg in cfg2 cfg3; do for rep in 1 2 3; do for v in 0 1; do
  env $VAR=$v python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs --no-live-pmc 2> gpurun_out/r5s_err.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('$cfg $VAR=$v', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"
done; done; done
