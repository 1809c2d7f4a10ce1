#!/bin/bash
# A/B on the GPU box: parity tests, env.step timing at two action scales, in-kernel phase profile (profile build)
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python tools/time_step.py --steps 40 --scale 0.3 > gpurun_out/ts.txt 2>&1 && python tools/time_step.py --steps 40 >> gpurun_out/ts.txt 2>&1; grep block gpurun_out/ts.txt
TMJX_SO=track_mjx_amd/libtmjx_hip_prof.so python tools/phase_profile.py > gpurun_out/phase.txt 2>&1; grep -v amdgpu gpurun_out/phase.txt
