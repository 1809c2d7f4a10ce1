#!/bin/bash
# K2 in one GPU call: in-kernel phase profile (TMW_PROFILE build), SQ counter passes incl. the LDS bank-conflict counters, instruction mix.
# usage (GPU box): bash tools/k2_profile.sh <tag>   -> gpurun_out/<tag>_phase_profile.txt, gpurun_out/sq_<tag>/, gpurun_out/pmc1_<tag>lds/
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TMJX_SO=track_mjx_amd/libtmjx_hip_prof.so timeout -k 10 300 python3 tools/phase_profile.py > gpurun_out/${TAG}_phase_profile.txt 2>&1
cat gpurun_out/${TAG}_phase_profile.txt
bash tools/sq_counters.sh $TAG
bash tools/pmc_one.sh ${TAG}lds k_physics_wave "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- python3 tools/time_step.py --steps 4 --scale 0.3 > gpurun_out/${TAG}_lds_counters.txt 2>&1
cat gpurun_out/${TAG}_lds_counters.txt
python3 tools/sq_summary.py gpurun_out/sq_$TAG ${2:-11} > gpurun_out/${TAG}_sq_summary.json 2>&1; tail -30 gpurun_out/${TAG}_sq_summary.json
