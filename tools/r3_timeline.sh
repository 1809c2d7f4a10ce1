#!/bin/bash
# timeline of one SGD minibatch step (cfg2) from a rocprofv3 kernel trace (rocpd output)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
DB=$(find $OUT/trace -name "*.db" | head -1)
echo "db=$DB"
python3 tools/step_timeline.py $DB 10 > $OUT/timeline.txt 2>&1
python3 tools/sgd_window.py $DB 40 > $OUT/window.txt 2>&1
tail -80 $OUT/timeline.txt
rm -rf $OUT/trace
