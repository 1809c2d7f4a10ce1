#!/bin/bash
# tools/r3_ab.sh <tag> <new.so> <old.so>: parity (strict + main GPU parity file) with the new build, then K2 A/B and bench A/B, same box
TAG=$1; NEW=$2; OLD=$3
mkdir -p gpurun_out/$TAG
TMJX_SO=$NEW python -m pytest tests/test_gpu_parity_strict.py tests/test_gpu_parity.py -x -q -m gpu -s > gpurun_out/$TAG/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|mean per solve" gpurun_out/$TAG/tests.log | cut -c1-330
rm -f gpurun_out/abk2_$TAG.txt gpurun_out/ab_$TAG.txt
bash tools/ab_k2.sh $TAG $OLD $NEW
bash tools/ab_so.sh $TAG $OLD $NEW
