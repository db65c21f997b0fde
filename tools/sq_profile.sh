#!/bin/bash
# SQ issue / wait counters of one command's kernels:  tools/sq_profile.sh <out-prefix> <python script + args ...>
# (run on the GPU box; two passes of 8 SQ counters each, summarised by tools/sq_ratios.py as fractions of SQ_WAVE_CYCLES)
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/${OUT}_a -- python3 "$@" > gpurun_out/${OUT}_a.log 2>&1
python tools/sq_ratios.py gpurun_out/${OUT}_a > gpurun_out/${OUT}_sq.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/${OUT}_b -- python3 "$@" > gpurun_out/${OUT}_b.log 2>&1
python tools/sq_ratios.py gpurun_out/${OUT}_b >> gpurun_out/${OUT}_sq.txt
rm -rf gpurun_out/${OUT}_a gpurun_out/${OUT}_b
