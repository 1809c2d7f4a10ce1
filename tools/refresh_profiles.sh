#!/bin/bash
# Round profiles in one GPU call: kernel-trace stats of the bench command, the GEMM micro-benchmark, the MFMA counters of the SGD half.
# usage (GPU box): bash tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>_*; copy what is to be judged into profiles/
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o ${TAG} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_trace.err
cp gpurun_out/${TAG}_trace/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null || find gpurun_out/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
python3 tools/gemm_bench.py cfg2 > gpurun_out/${TAG}_gemm_bench.txt 2>/dev/null && python3 tools/gemm_bench.py cfg4 >> gpurun_out/${TAG}_gemm_bench.txt 2>/dev/null
bash tools/mfma_counters.sh ${TAG} && python3 tools/mfma_summary.py gpurun_out/mfma_${TAG}
ls gpurun_out | grep ${TAG}
