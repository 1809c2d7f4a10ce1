#!/bin/bash
# rocprofv3 kernel stats of the cfg4 and cfg5 bench commands (GPU box) -> gpurun_out/cfgstats/{cfg4,cfg5}_kernel_stats.csv
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/cfgstats; mkdir -p $OUT
for CFG in cfg5 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$CFG -o x -- python3 bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only --no-other-configs > $OUT/bench_$CFG.json 2> $OUT/err_$CFG.txt
  echo "$CFG rc=$?"
  find $OUT/trace_$CFG -name "*kernel_stats.csv" -exec cp {} $OUT/${CFG}_kernel_stats.csv \;
  rm -rf $OUT/trace_$CFG
  head -8 $OUT/${CFG}_kernel_stats.csv | cut -c1-140
  echo "Cijk rows: $(grep -c "Cijk" $OUT/${CFG}_kernel_stats.csv)"
done
