#!/bin/bash
# MFMA counters of the learner's GEMM kernels over the SGD half of a training step (tools/sgd_step.py, eager launches) -> gpurun_out/mfma_<tag>;
# summarise with tools/mfma_summary.py gpurun_out/mfma_<tag>  (writes profiles/mfma_counters.json)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mfma_$1
mkdir -p $OUT
python3 tools/buildid.py --stamp $OUT > /dev/null
for c in cfg2 cfg4 cfg5; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/$c -- python3 tools/sgd_step.py --config $c > $OUT/$c.log 2>&1
grep config= $OUT/$c.log
done
