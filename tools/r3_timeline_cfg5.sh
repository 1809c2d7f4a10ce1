#!/bin/bash
# timeline of one SGD minibatch step of config 5 (bf16 GEMM inputs) from a rocprofv3 kernel trace (rocpd output)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl5
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -o x -- python3 bench.py --config cfg5 --steps 1 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 tools/step_timeline.py $DB 10 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
rm -rf $OUT/trace
