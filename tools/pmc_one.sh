#!/bin/bash
# rocprofv3 --pmc passes over one python command; prints per-kernel means.  usage: tools/pmc_one.sh <tag> <kernel substring> "<counters pass 1>" "<counters pass 2>" -- python3 script args
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; FILT=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
OUT=gpurun_out/pmc1_$TAG
mkdir -p $OUT
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$OUT" "$FILT" <<'PY'
import collections, csv, glob, sys
out, filt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            per[(r["Kernel_Name"][:50], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        agg[k][c].append(v)
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        xs = xs[len(xs) // 2:]
        print(f"   {c:32s} n={len(xs):4d} mean={sum(xs)/len(xs):16.1f}")
PY
