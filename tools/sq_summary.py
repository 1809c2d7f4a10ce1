#!/usr/bin/env python3
"""profiles/sq_counters.json from the two SQ passes of tools/sq_counters.sh: per wave-substep instruction counts of k_physics_wave, how
busy the vector pipe is and how many lanes an average vector instruction uses.

  valu_pipe_busy = SQ_INSTS_VALU x 2 cycles (a SIMD-32 executes a wave64 VALU instruction in 2 cycles, MI355X_MICROARCH.md) x waves
                   per SIMD / cycles a wave is resident (SQ_WAVE_CYCLES x 4: the counter ticks every 4 clocks)
  lane_occupancy = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)   (the gfx9 VALUUtilization formula)
usage: python tools/sq_summary.py gpurun_out/sq_<tag> [envs_per_cu]"""
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
_pos = [a for a in sys.argv[2:] if not a.startswith("--")]
envs_per_cu = float(_pos[0]) if _pos else 11.0
ENVS, NSUB = 4096, 10
tot = collections.defaultdict(list)
for sub in ("a", "b"):
    for f in glob.glob(str(src / sub / "*" / "*_counter_collection.csv")):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void k_physics_wave"):
                per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(per, key=int)
        for d in (ids[3:] if len(ids) > 4 else ids):      # steady-state env.step launches (the first ones: reset forward / warm-up)
            for k, v in per[d].items():
                tot[k].append(v)
c = {k: sum(v) / len(v) / (ENVS * NSUB) for k, v in tot.items()}        # per wave (= env) and substep
waves_per_simd = envs_per_cu / 4.0
res = {"note": "rocprofv3 --pmc SQ_* (two passes, tools/sq_counters.sh) over tools/time_step.py --steps 4 --scale 0.3, k_physics_wave<true>, 4096 envs in one launch; "
               "values per wave-substep; SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* tick every 4 clocks",
       "envs_per_cu": envs_per_cu, "waves_per_simd": waves_per_simd, "per_wave_substep": {k: round(v, 1) for k, v in sorted(c.items())}}
if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c:
    res["SQ_INSTS_VALU_per_wave_substep"] = round(c["SQ_INSTS_VALU"], 1)
    res["wave_resident_cycles_per_substep"] = round(4 * c["SQ_WAVE_CYCLES"], 0)
    res["valu_pipe_busy"] = round(2.0 * c["SQ_INSTS_VALU"] * waves_per_simd / (4.0 * c["SQ_WAVE_CYCLES"]), 4)
    res["wave_issues_valu_fraction_of_its_residency"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"], 4)
if c.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in c:
    res["lane_occupancy"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    res["wave_parked_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    res["issue_stall_fraction"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
res.update(BUILD)
(Path(__file__).resolve().parents[1] / "profiles" / "sq_counters.json").write_text(json.dumps(res, indent=1) + "\n")
print(json.dumps(res, indent=1))
