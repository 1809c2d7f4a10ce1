#!/bin/bash
# round 5, call 8: SGD experiments — one stream against two, bf16 weight-gradient slab counts, cfg3 with the 1024-workgroup dW group
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
for rep in 1 2; do
echo "cfg2 two streams: $(timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1)"
echo "cfg2 one stream:  $(TMJX_SGD_TWO_STREAMS=0 timeout -k 10 120 python tools/sgd_step.py --config cfg2 --graph --updates 4 2>&1 | tail -1)"
done
echo "cfg3 two streams: $(timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1)"
echo "cfg3 one stream:  $(TMJX_SGD_TWO_STREAMS=0 timeout -k 10 120 python tools/sgd_step.py --config cfg3 --graph --updates 4 2>&1 | tail -1)"
echo "cfg4 two streams: $(timeout -k 10 120 python tools/sgd_step.py --config cfg4 --graph --updates 2 2>&1 | tail -1)"
echo "cfg4 one stream:  $(TMJX_SGD_TWO_STREAMS=0 timeout -k 10 120 python tools/sgd_step.py --config cfg4 --graph --updates 2 2>&1 | tail -1)"
for W in 512 256 128 1024; do echo "cfg5 TMJX_BDW_WGS=$W: $(TMJX_BDW_WGS=$W timeout -k 10 150 python tools/sgd_step.py --config cfg5 --graph --updates 2 2>&1 | tail -1)"; done
echo "cfg5 one stream: $(TMJX_SGD_TWO_STREAMS=0 timeout -k 10 150 python tools/sgd_step.py --config cfg5 --graph --updates 2 2>&1 | tail -1)"
