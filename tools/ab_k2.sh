#!/bin/bash
# A/B of library builds on the physics kernel alone: env.step of one launch of 4096 envs at action scale 0.3 (REPS alternating runs of 40 steps) per build
TAG=$1; shift
REPS=${REPS:-4}
mkdir -p gpurun_out
for rep in $(seq 1 $REPS); do
for so in "$@"; do
  r=$(TMJX_SO=$so python tools/time_step.py --steps 40 --scale 0.3 2>&1 | grep block | sed 's/.*ms\/step=\([0-9.]*\).*/\1/')
  echo "$so rep$rep ms/step=$r" >> gpurun_out/abk2_$TAG.txt
done
done
sort gpurun_out/abk2_$TAG.txt
