#!/bin/bash
# VALU / SALU / LDS instruction counts of the physics kernel by phase: builds that leave the substep after phase TMW_STOP (results are
# garbage, the counts are additive); the Euler step still runs in every variant
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1 2 3 4; do
  TMJX_SO=track_mjx_amd/libtmjx_stop$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/vp_$v -- python3 tools/scratch/valu_count.py 5 5 > gpurun_out/vp_$v.log 2>&1
done
python3 - <<'P'
import csv,glob,collections
for tag in ("0","1","2","3","4"):
    f=glob.glob(f"gpurun_out/vp_{tag}/*/*_counter_collection.csv")
    if not f: print(tag,"missing"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_physics_wave" in r["Kernel_Name"] and int(r["Grid_Size"])==4096*64: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("stop after phase", tag, {k: round(sum(v[2:])/len(v[2:])/40960,1) for k,v in agg.items()})
P
