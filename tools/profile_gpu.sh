#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command, then two separate PMC passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") over a short rollout.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$1
mkdir -p $OUT
python3 tools/buildid.py --stamp $OUT > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --no-rollout-only --no-cpu-baseline --no-other-configs > $OUT/bench.log 2>&1
grep '^{"metric' $OUT/bench.log | tail -1 > $OUT/bench.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/time_step.py --steps 4 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/time_step.py --steps 4 > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -40
