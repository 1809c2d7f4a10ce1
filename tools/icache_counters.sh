#!/bin/bash
# instruction-cache counters of the physics kernel (rocprofv3 --pmc, counters + kernel trace only): bash tools/icache_counters.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/icache_$1
mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- python3 tools/time_step.py --steps 4 --scale 0.3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQC_TC_INST_REQ SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 tools/time_step.py --steps 4 --scale 0.3 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for run in "ab":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True)
    if not f: print("no counters for pass", run); continue
    acc = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f[0])):
        if "k_physics_wave" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    disp = sum(1 for r in csv.DictReader(open(f[0])) if "k_physics_wave" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_WAVE_CYCLES")
    print("pass", run, "dispatches", disp, {k: round(v / max(disp, 1)) for k, v in acc.items()})
PY
