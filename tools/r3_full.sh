#!/bin/bash
# full GPU suite + headline bench + rocprof kernel stats.  usage: bash tools/r3_full.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/full_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; RC=$?; echo "tests rc=$RC"; tail -5 $OUT/tests.log
[ $RC -eq 0 ] || exit 1
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
o=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c=o["config"]
print(round(o["value"]), round(o["ms_per_step"],2), "rollout", round(c["rollout_ms_per_step"],2), "sgd", round(c["sgd_ms_per_step"],2), "per-mb", round(c["sgd_ms_per_minibatch_step"],4), "rollout-only", round(c["rollout_only_env_steps_per_s_per_gpu"]), "cpu", o.get("cpu_baseline",{}).get("value"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r03 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-rollout-only > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
head -12 $OUT/kernel_stats.csv | cut -c1-150
grep -c "at::native" $OUT/kernel_stats.csv
