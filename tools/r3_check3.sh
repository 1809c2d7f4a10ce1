#!/bin/bash
set -u
mkdir -p gpurun_out/r3e
echo skip tests
T0=$(date +%s); python bench.py > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
o=json.loads([l for l in open("gpurun_out/r3e/bench.json") if l.startswith("{")][-1])
print(o["value"], o["ms_per_step"]); print(json.dumps(o["config"].get("other_configs"), indent=1)[:1500]); print(o["cpu_baseline"])
PY
