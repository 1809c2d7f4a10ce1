#!/bin/bash
# extra bench lines (BASELINE.json configs[3], configs[4]), SGD-step timings and the MFMA counters of the SGD step
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cfgs
python bench.py --config cfg4 --no-cpu-baseline > gpurun_out/cfgs/bench_cfg4.json 2> gpurun_out/cfgs/bench_cfg4.err; cut -c1-200 gpurun_out/cfgs/bench_cfg4.json
python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/cfgs/bench_cfg5.json 2> gpurun_out/cfgs/bench_cfg5.err; cut -c1-200 gpurun_out/cfgs/bench_cfg5.json
for c in cfg2 cfg4 cfg5; do python tools/sgd_step.py --config $c --graph 2>&1 | grep config=; done | tee gpurun_out/cfgs/sgd_step.txt
for c in cfg2 cfg5; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d gpurun_out/cfgs/pmc_mfma_$c -- python3 tools/sgd_step.py --config $c > gpurun_out/cfgs/pmc_mfma_$c.log 2>&1
done
