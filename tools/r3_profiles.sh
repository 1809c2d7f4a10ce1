#!/bin/bash
# round-3 counter refresh: HBM traffic (FETCH / WRITE passes), SQ counters of the physics kernel, MFMA counters of the GEMM kernels (cfg2 / cfg4 / cfg5)
set -u
bash tools/profile_gpu.sh r03 > gpurun_out/prof_r03.log 2>&1; echo "profile_gpu rc=$?"
bash tools/sq_counters.sh r03 > gpurun_out/sq_r03.log 2>&1; echo "sq rc=$?"
bash tools/mfma_counters.sh r03 > gpurun_out/mfma_r03.log 2>&1; echo "mfma rc=$?"
tail -3 gpurun_out/mfma_r03.log
du -sh gpurun_out/prof_r03 gpurun_out/sq_r03 gpurun_out/mfma_r03
