#!/bin/bash
# round 5, call 15: whole GPU suite on the product build; the bf16 tests and config 5 under the fp32-pre-activation build (-DTMJX_BF16_Z_F32): both variants, once
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
TMJX_SO=alt/libtmjx_zf32.so timeout -k 10 600 python -m pytest tests/test_gpu_gemm_bf16.py tests/test_gpu_parity.py -m gpu -q -k "bf16 or bgemm or shadow" > $O/tests_zf32.txt 2>&1; echo "fp32-z build bf16 tests rc=$?"; tail -3 $O/tests_zf32.txt
for rep in 1 2; do
echo "cfg5 SGD, bf16 z (product):  $(timeout -k 10 150 python tools/sgd_step.py --config cfg5 --graph --updates 2 2>&1 | tail -1)"
echo "cfg5 SGD, fp32 z (-DTMJX_BF16_Z_F32): $(TMJX_SO=alt/libtmjx_zf32.so timeout -k 10 150 python tools/sgd_step.py --config cfg5 --graph --updates 2 2>&1 | tail -1)"
done | tee $O/cfg5_z_variants.txt
for v in track_mjx_amd/libtmjx_hip.so alt/libtmjx_zf32.so; do TMJX_SO=$v python bench.py --config cfg5 --steps 4 --warmup 2 --no-cpu-baseline --no-rollout-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('cfg5 bench $v: value %.0f  rollout_ms %.1f sgd_ms/minibatch %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_minibatch_step']))"; done | tee -a $O/cfg5_z_variants.txt
python bench.py --config cfg4 --steps 4 --warmup 2 --no-cpu-baseline --no-rollout-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('cfg4 bench: value %.0f  rollout_ms %.1f sgd_ms/minibatch %.3f' % (d['value'], c['rollout_ms_per_step'], c['sgd_ms_per_minibatch_step']))"
