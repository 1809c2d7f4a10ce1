cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02c -o r02c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only > gpurun_out/prof_r02c_bench.json 2> gpurun_out/prof_r02c.err
python tools/sgd_window.py gpurun_out/prof_r02c/r02c_results.db 45
