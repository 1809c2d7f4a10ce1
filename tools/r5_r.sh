#!/bin/bash
# round 5, call r: 32-row GEMM tiles forced at config 2's 20 480 rows (the tile chooser's cost model counts lock-step rounds of 256 workgroups; 256 tiles of 80 rows are
# ONE workgroup = one wave per SIMD on every CU) against the chooser's pick
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do for mt in 0 2; do
  TMJX_GEMM_MT=$mt python bench.py --config cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs --no-live-pmc 2> gpurun_out/r5r_err.txt | grep '^{' | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print('cfg2 TMJX_GEMM_MT=$mt', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1))"
done; done
