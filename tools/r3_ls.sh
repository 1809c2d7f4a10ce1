#!/bin/bash
# line search: unfused derivative (ls_new) + exact-zero shortcuts (ls_short): parity with the newest build, then K2 A/B and bench A/B
mkdir -p gpurun_out/ls
TMJX_SO=build_ab/ls_short.so python -m pytest tests/test_gpu_parity_strict.py tests/test_gpu_parity.py -x -q -m gpu -s > gpurun_out/ls/tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|mean per solve" gpurun_out/ls/tests.log | cut -c1-400
rm -f gpurun_out/abk2_ls.txt gpurun_out/ab_ls.txt
bash tools/ab_k2.sh ls build_ab/jt_old.so build_ab/ls_new.so build_ab/ls_short.so
bash tools/ab_so.sh ls build_ab/ls_new.so build_ab/ls_short.so
