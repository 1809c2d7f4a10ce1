#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "5 5" "1 5" "5 1" "0 0"; do
  set -- $cfg
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/vc_$1_$2 -- python3 tools/scratch/valu_count.py $1 $2 > gpurun_out/vc_$1_$2.log 2>&1
done
python3 - <<'P'
import csv,glob,collections
for tag in ("5_5","1_5","5_1","0_0"):
    f=glob.glob(f"gpurun_out/vc_{tag}/*/*_counter_collection.csv")
    if not f: print(tag,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_physics_wave" in r["Kernel_Name"] and int(r["Grid_Size"])==4096*64: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v[2:])/len(v[2:])/40960,1) for k,v in agg.items()})
P
