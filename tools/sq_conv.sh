#!/bin/bash
# SQ counters of single conv calls at the bench layer shapes:  tools/sq_conv.sh <tag> [conv_bench.py args ...]
#   -> gpurun_out/<tag>_sq.txt (MFMA busy, VALU per MFMA, LDS bank conflicts per kernel instantiation) + the issue / wait view
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/${TAG}_a -- python3 tools/conv_bench.py "$@" --reps 3 > gpurun_out/${TAG}_a.log 2>&1
python profiles/summarise_sq.py gpurun_out/${TAG}_a "rocprofv3 --pmc SQ_* -- python3 tools/conv_bench.py $* --reps 3" > gpurun_out/${TAG}_sq.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/${TAG}_b -- python3 tools/conv_bench.py "$@" --reps 3 > gpurun_out/${TAG}_b.log 2>&1
python tools/sq_ratios.py gpurun_out/${TAG}_b >> gpurun_out/${TAG}_sq.txt
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b
