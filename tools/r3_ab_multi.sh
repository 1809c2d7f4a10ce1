#!/bin/bash
# tools/r3_ab_multi.sh <tag> <old.so> <new1.so> [<new2.so> ...]: strict + main parity files with every new build, then the K2 A/B of all builds on one box
TAG=$1; OLD=$2; shift 2
mkdir -p gpurun_out/$TAG
for NEW in "$@"; do
  b=$(basename $NEW .so)
  TMJX_SO=$NEW timeout -k 10 400 python -m pytest tests/test_gpu_parity_strict.py tests/test_gpu_parity.py -x -q -m gpu -s > gpurun_out/$TAG/tests_$b.log 2>&1
  echo "$b tests rc=$?"; grep -E "passed|failed|mean per solve" gpurun_out/$TAG/tests_$b.log | cut -c1-300
done
rm -f gpurun_out/abk2_$TAG.txt
REPS=${REPS:-3} bash tools/ab_k2.sh $TAG $OLD "$@"
