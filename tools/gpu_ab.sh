#!/bin/bash
# One parameterised GPU-box script for "parity with the new build, then A/B against an older build on the same box" (replaces the per-experiment
# tools/r3_*.sh lab scripts of round 3).
#   usage: bash tools/gpu_ab.sh <tag> [-t "<pytest files>"] [-k "<pytest -k expr>"] [-o old.so] [-n new.so] [-b] [-r reps] [-m "<micro binaries>"]
#   -t  test files run with the new build (default: tests/test_gpu_parity_strict.py tests/test_gpu_parity.py); "none" skips
#   -o / -n  libraries for the A/B legs (default new: the in-tree track_mjx_amd/libtmjx_hip.so); without -o no A/B
#   -b  also A/B the bench line (tools/ab_so.sh) after the env.step A/B (tools/ab_k2.sh)
# everything lands in gpurun_out/<tag>/
set -u
TAG=$1; shift
TESTS="tests/test_gpu_parity_strict.py tests/test_gpu_parity.py"; KEXPR=""; OLD=""; NEW="track_mjx_amd/libtmjx_hip.so"; BENCH=0; MICRO=""
export REPS=${REPS:-4}
while getopts "t:k:o:n:br:m:" opt; do
  case $opt in
    t) TESTS=$OPTARG;; k) KEXPR=$OPTARG;; o) OLD=$OPTARG;; n) NEW=$OPTARG;; b) BENCH=1;; r) export REPS=$OPTARG;; m) MICRO=$OPTARG;;
  esac
done
OUT=gpurun_out/$TAG; mkdir -p $OUT
for m in $MICRO; do echo "== $m" >> $OUT/micro.txt; timeout -k 10 120 $m >> $OUT/micro.txt 2>&1; done
[ -n "$MICRO" ] && cat $OUT/micro.txt
if [ "$TESTS" != "none" ]; then
  TMJX_SO=$NEW timeout -k 10 900 python -m pytest $TESTS -x -q -m gpu -s ${KEXPR:+-k "$KEXPR"} > $OUT/tests.log 2>&1
  echo "tests rc=$?"; grep -E "passed|failed|^\[scale|median|mean per solve|cfg1|K3:|Error" $OUT/tests.log | cut -c1-400
fi
if [ -n "$OLD" ]; then
  rm -f gpurun_out/abk2_$TAG.txt gpurun_out/ab_$TAG.txt
  bash tools/ab_k2.sh $TAG $OLD $NEW && cp gpurun_out/abk2_$TAG.txt $OUT/
  [ $BENCH = 1 ] && bash tools/ab_so.sh $TAG $OLD $NEW && cp gpurun_out/ab_$TAG.txt $OUT/
fi
true
