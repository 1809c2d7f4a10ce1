#!/bin/bash
# SQ counter passes over the physics kernel (rocprofv3 --pmc, 8 SQ slots per pass, counters only: no trace domains besides the kernel
# trace), workload = tools/time_step.py --steps 4 --scale 0.3 at 4096 envs.  Summarise with tools/sq_summary.py gpurun_out/sq_<tag>.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$1
mkdir -p $OUT
python3 tools/buildid.py --stamp $OUT > /dev/null
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a -- python3 tools/time_step.py --steps 4 --scale 0.3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- python3 tools/time_step.py --steps 4 --scale 0.3 > $OUT/b.log 2>&1
tail -n 2 $OUT/a.log $OUT/b.log
