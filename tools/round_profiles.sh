#!/bin/bash
# Everything the round's DESIGN / bench figures are read from, in ONE GPU call (≈ 10 min): bash tools/round_profiles.sh <tag> [envs per CU]
#   -> gpurun_out/<tag>_*: default bench line (+ cfg4 / cfg5 inside it), kernel stats of the bench command, PMC traffic, SQ counters (+ LDS bank
#      conflicts), MFMA counters of the SGD step, in-kernel phase profile, strict parity log, kernel stats of the cfg4 / cfg5 bench commands
# afterwards on the build machine: python tools/pmc_summary.py gpurun_out/prof_<tag>; python tools/sq_summary.py gpurun_out/sq_<tag> <envs/CU>;
#   python tools/mfma_summary.py gpurun_out/mfma_<tag>   (they write profiles/*.json, stamped with the build id), and copy the rest into profiles/
set -u
TAG=$1; EPC=${2:-12}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo "== bench (default command)"; python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
echo "== strict parity"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity_strict.py tests/test_gpu_parity.py -q -m gpu -s -k "strict or worst_env or config1 or full_size or substeps_teacher" > gpurun_out/${TAG}_parity_strict.log 2>&1; tail -2 gpurun_out/${TAG}_parity_strict.log
echo "== kernel stats + PMC traffic"; bash tools/profile_gpu.sh $TAG > gpurun_out/${TAG}_profile_gpu.log 2>&1; find gpurun_out/prof_$TAG/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \; ; head -6 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
echo "== K2 counters"; bash tools/k2_profile.sh $TAG $EPC > gpurun_out/${TAG}_k2_profile.log 2>&1; tail -12 gpurun_out/${TAG}_k2_profile.log
echo "== MFMA counters"; bash tools/mfma_counters.sh $TAG
echo "== cfg4 / cfg5 kernel stats"; bash tools/gpu_lab.sh cfgstats > gpurun_out/${TAG}_cfgstats.log 2>&1; grep -E "rc=|Cijk" gpurun_out/${TAG}_cfgstats.log
echo "== roll-out timeline"; bash tools/gpu_lab.sh timeline cfg2 > gpurun_out/${TAG}_timeline.log 2>&1; grep -E "serial phase|median" gpurun_out/${TAG}_timeline.log
