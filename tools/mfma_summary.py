#!/usr/bin/env python3
"""profiles/mfma_counters.json from tools/mfma_counters.sh: matrix-pipe utilisation inside the learner's GEMM kernels (k_gemm_act / k_gemm_dw)
over the SGD half of a training step.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) (gfx94x MfmaUtil formula,
per-kernel sums).  usage: python tools/mfma_summary.py gpurun_out/mfma_<tag>"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

src = Path(sys.argv[1])
sys.path.insert(0, str(Path(__file__).resolve().parent))
from buildid import checked_id  # noqa: E402
BUILD = checked_id(src, "--force" in sys.argv)
res = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA ... over tools/sgd_step.py --config <cfg> (eager launches); "
               "utilisation = MFMA busy cycles / (4 x CU busy cycles), summed per kernel family"}
for cfg in ("cfg2", "cfg4", "cfg5"):
    fs = glob.glob(str(src / cfg / "*" / "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        fam = ("k_chain_fwd / k_chain_bwd (whole chains, 256-wide nets)" if "k_chain_" in k else
               "k_gemm_act (forward / input gradient)" if "k_gemm_act" in k else "k_gemm_dw (weight gradient)" if "k_gemm_dw" in k else
               "k_bgemm_nt (bf16 forward / input gradient, fused epilogues)" if "k_bgemm_nt" in k else "k_bgemm_dw (bf16 weight gradient)" if "k_bgemm_dw<" in k else
               ("library GEMM" if k.startswith("Cijk") else None))
        if fam:
            agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    out, tot = {}, collections.defaultdict(float)
    for fam, v in agg.items():
        out[fam] = {"mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * v["SQ_BUSY_CU_CYCLES"], 1), 4), "mfma_insts": v["SQ_INSTS_MFMA"],
                    "wave_parked_fraction": round(v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 4),
                    "issue_stall_fraction": round(v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), 4)}
        if not fam.startswith("library"):
            for n, x in v.items():
                tot[n] += x
    out["all_hand_written_gemm_kernels"] = round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * tot["SQ_BUSY_CU_CYCLES"], 1), 4)
    out["library_gemm_kernels_in_timed_region"] = int(any(f.startswith("library") for f in agg))
    res[cfg] = out
res["mfma_util"] = res.get("cfg2", {}).get("all_hand_written_gemm_kernels")
res.update(BUILD)
(Path(__file__).resolve().parents[1] / "profiles" / "mfma_counters.json").write_text(json.dumps(res, indent=1) + "\n")
print(json.dumps(res, indent=1))
