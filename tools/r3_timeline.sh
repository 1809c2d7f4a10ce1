#!/bin/bash
# timelines from a rocprofv3 kernel trace (rocpd output) of the bench: one SGD minibatch step, one env group's serial roll-out phase
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/trace -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
DB=$(find $OUT/trace -name "*.db" | head -1)
python3 tools/step_timeline.py $DB 10 > $OUT/timeline.txt 2>&1
python3 tools/rollout_timeline.py $DB 200 > $OUT/rollout_timeline.txt 2>&1
cat $OUT/rollout_timeline.txt
rm -rf $OUT/trace
