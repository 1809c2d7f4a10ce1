#!/bin/bash
# occupancy sensitivity of the physics kernel: pad the dynamic LDS request so that fewer envs fit a CU
python tools/time_step.py --steps 20 --scale 0.3 > /dev/null 2>&1
for pad in 0 1 4 6 10; do
  echo "TMJX_LDS_PAD_KB=$pad"; TMJX_LDS_PAD_KB=$pad python tools/time_step.py --steps 60 --scale 0.3 2>&1 | grep block
done
