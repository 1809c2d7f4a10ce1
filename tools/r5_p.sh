#!/bin/bash
# round 5, call 17: multi-rank rehearsal on the one GPU (gloo ranks sharing cuda:0: plumbing only) incl. the cfg3 line, then the round's profile collection
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
bash tools/gpu_lab.sh rehearse > $O/rehearse.txt 2>&1; cat $O/rehearse.txt | tail -12
TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --config cfg3 --envs-per-gpu 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-rollout-only > $O/cfg3_n2.json 2> $O/cfg3_n2.err
echo "cfg3 N=2 rc=$?"; grep '^{' $O/cfg3_n2.json | tail -1 | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; print(o['n_gpus'], round(o['value']), c['ranks_seen'], c['parallelism'], c['global_batch'], c['minibatch_gemm_rows'], c['collectives'], c.get('rehearsal','')[:30])"
