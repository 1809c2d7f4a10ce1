#!/bin/bash
# round 5, call x: the multi-rank paths on the final build: (1) config 5 with a one-rank RCCL group (17 MB of gradients: the bucketed three-graph step, bf16 chain, grouped weight
# gradients per bucket) against the plain run; (2) the 2 / 4-rank rehearsal over gloo on one GPU (tools/gpu_lab.sh rehearse)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=o['config']; print('$1', round(o['value']), 'sgd ms', round(c['sgd_ms_per_minibatch_step'],4), 'rollout ms', round(c['rollout_ms_per_step'],1), 'collectives', c.get('collectives'), 'ranks', c.get('ranks_seen'))"; }
python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-live-pmc --no-rollout-only 2> gpurun_out/r5x_plain.err | grep '^{' | line "cfg5 plain"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 TMJX_COLLECTIVES_ALWAYS=1 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-live-pmc --no-rollout-only 2> gpurun_out/r5x_rccl.err | grep '^{' | line "cfg5 one-rank RCCL"
grep -i "capture\|eager\|error" gpurun_out/r5x_rccl.err | head -5
bash tools/gpu_lab.sh rehearse
