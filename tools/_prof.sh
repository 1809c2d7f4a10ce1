cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02b -o r02b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rollout-only > gpurun_out/prof_r02b_bench.json 2> gpurun_out/prof_r02b.err
ls gpurun_out/prof_r02b | head
