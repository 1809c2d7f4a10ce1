#!/bin/bash
# rocprofv3 --pmc <counters...> over a python script; prints per-kernel averages of each counter for kernels matching a filter
# usage: tools/pmc_kernel.sh <outtag> <kernel substring> <script.py> <counters...>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; FILT=$2; SCRIPT=$3; shift 3
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $SCRIPT > $OUT/log.txt 2>&1
python3 - "$OUT" "$FILT" <<'PY'
import collections, csv, glob, sys
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:32s} n={len(xs):4d} mean={sum(xs)/len(xs):14.1f}")
PY
