cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02g -o r02g -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only > gpurun_out/prof_r02g_bench.json 2> gpurun_out/prof_r02g.err
python tools/step_timeline.py gpurun_out/prof_r02g/r02g_results.db 10
