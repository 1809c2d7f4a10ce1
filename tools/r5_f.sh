#!/bin/bash
# round 5, call 6: which of the round's LDS moves makes the action-repeat twin test differ on the GPU (bit-identical in the host emulation)?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
for v in alt/libtmjx_lb3.so alt/libtmjx_va.so alt/libtmjx_vb.so alt/libtmjx_vc.so alt/libtmjx_vd.so; do
  TMJX_SO=$v timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "action_repeat" > $O/t_$(basename $v).txt 2>&1; echo "$v action_repeat rc=$?"
done
# the whole parity suite (no -x) under the 3-wave build that fails deterministically, and under the product build
TMJX_SO=alt/libtmjx_lb3.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_strict.py -m gpu -q > $O/all_lb3.txt 2>&1; echo "lb3 all rc=$?"; tail -8 $O/all_lb3.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_strict.py -m gpu -q > $O/all_cur.txt 2>&1; echo "cur all rc=$?"; tail -8 $O/all_cur.txt
