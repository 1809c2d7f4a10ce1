cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg4 -o cfg4 -- python3 bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-rollout-only > gpurun_out/prof_cfg4_bench.json 2> gpurun_out/prof_cfg4.err
python tools/sgd_window.py gpurun_out/prof_cfg4/cfg4_results.db 30
