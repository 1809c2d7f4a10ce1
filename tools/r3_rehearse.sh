#!/bin/bash
# two and four ranks of bench.py on ONE GPU (gloo collectives on device tensors): the N > 1 path of the bench end to end.  GPU box.
set -u
mkdir -p gpurun_out/rehearse
for N in 2 4; do
  TMJX_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
    bench.py --gpus $N --envs-per-gpu 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-rollout-only > gpurun_out/rehearse/n$N.json 2> gpurun_out/rehearse/n$N.err
  echo "N=$N rc=$?"
  python - gpurun_out/rehearse/n$N.json <<'PY'
import json, sys
ls=[l for l in open(sys.argv[1]) if l.startswith("{")]
print(len(ls), "JSON line(s)")
if ls:
    o=json.loads(ls[-1]); c=o["config"]
    print(o["n_gpus"], round(o["value"]), c["ranks_seen"], c["parallelism"], c["global_batch"], c["env_steps_per_step"], c.get("rehearsal","")[:40])
PY
  tail -3 gpurun_out/rehearse/n$N.err
done
for N in 2 3; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29530 + N)) tools/two_rank_sync_check.py 3 > gpurun_out/rehearse/sync$N.txt 2> gpurun_out/rehearse/sync$N.err
  echo "sync N=$N rc=$?"; grep "ranks on one GPU" gpurun_out/rehearse/sync$N.txt; grep -i "error\|assert" gpurun_out/rehearse/sync$N.err | head -5
done
