#!/usr/bin/env python3
"""profiles/r01_mfma_counters.txt from the PMC runs of tools/run_cfgs.sh (MFMA counters of the SGD half of a training step).
Usage: python tools/mfma_summary.py gpurun_out/cfgs  > profiles/r01_mfma_counters.txt"""
import collections
import csv
import glob
import sys

src = sys.argv[1]
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace")
print("  -- python3 tools/sgd_step.py --config <cfg>   (the SGD half of a training step, eager launches; tools/run_cfgs.sh)")
print("MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  (gfx94x MfmaUtil formula, per-kernel sums)\n")
for c, label in (("cfg2", "2x256 nets, fp32 GEMMs (hipBLASLt v_mfma_f32_16x16x4_f32)"), ("cfg5", "rodent-mc-intention nets, bf16 GEMM inputs / fp32 accumulate and output (agent/networks.py: gemm_inputs)")):
    fs = glob.glob(f"{src}/pmc_mfma_{c}/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:72]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_MFMA":
            calls[k] += 1
    print(f"== {c}: {label}")
    tot = collections.defaultdict(float)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        if not k.startswith("Cijk"):
            continue
        for n, x in v.items():
            tot[n] += x
        print(f"  {k:72s} calls={calls[k]:5d} mfma_util={v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(4 * v['SQ_BUSY_CU_CYCLES'], 1):.3f} mfma_insts={v['SQ_INSTS_MFMA']:.3g}")
    print(f"  all library GEMM kernels: MFMA utilisation {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(4 * tot['SQ_BUSY_CU_CYCLES'], 1):.3f}  (MFMA instructions {tot['SQ_INSTS_MFMA']:.4g}, MOPS f32 {tot['SQ_INSTS_VALU_MFMA_MOPS_F32']:.4g}, bf16 {tot['SQ_INSTS_VALU_MFMA_MOPS_BF16']:.4g})\n")
