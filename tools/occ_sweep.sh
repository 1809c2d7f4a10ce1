#!/bin/bash
# residency sensitivity of the physics kernel WITHOUT launch quantisation: the env count of each run is exactly what fits the chip at that
# residency (256 CUs x envs per CU; the LDS pad sets the envs per CU: 0/1/2/3/6 KB -> 12/11/10/9/8 per CU at 12 780 bytes per env)
python tools/time_step.py --steps 20 --scale 0.3 > /dev/null 2>&1
for so in "$@"; do
for cfg in "0 3072" "1 2816" "2 2560" "6 2048" "6 1024" "0 6144" "6 4096"; do
  set -- $cfg
  echo "$so pad=$1 $(TMJX_SO=$so TMJX_LDS_PAD_KB=$1 python tools/time_step.py --envs $2 --steps 60 --scale 0.3 2>&1 | grep block)"
done; done
