#!/bin/bash
# One call on the GPU box: kernel trace, the two PMC traffic passes and the SQ pass of the bench step, both precisions.
#   bash tools/profile_round.sh r03        -> gpurun_out/ck/r03_* (copy what is to be judged into profiles/)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ck
FLAGS="--no-cpu-baseline --no-extra --no-instrumented"
# kernel durations and counters are taken with every pass of the step on ONE stream (a kernel alone on the chip, as bench.py's
# instrumented pass measures them); the headline step runs the independent passes on forked streams
export AVSEP_FORK_SOURCES=0 AVSEP_FORK_PAIR=0
python bench.py --steps 2 --warmup 1 $FLAGS > gpurun_out/ck/warm.log 2>&1
for P in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ck/tr_$P -- python3 bench.py --steps 3 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/tr_$P.log 2>&1
  python profiles/summarise_trace.py gpurun_out/ck/tr_$P "python3 bench.py --steps 3 --warmup 1 $FLAGS --precision $P (full-HIP AV step, batch 64, 4 steps in the trace; AVSEP_FORK_SOURCES=0 AVSEP_FORK_PAIR=0: one stream)" > gpurun_out/ck/${TAG}_bench_kernel_stats_$P.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/ck/f_$P -- python3 bench.py --steps 2 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/f_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/ck/w_$P -- python3 bench.py --steps 2 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/w_$P.log 2>&1
  python profiles/summarise_pmc.py gpurun_out/ck/f_$P gpurun_out/ck/w_$P "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 $FLAGS --precision $P" 3 64 gpurun_out/ck/pmc_$P.json > gpurun_out/ck/pmc_$P.log
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/ck/sq_$P -- python3 bench.py --steps 1 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/sq_$P.log 2>&1
  python profiles/summarise_sq.py gpurun_out/ck/sq_$P "rocprofv3 --pmc SQ_* --kernel-trace -- python3 bench.py --steps 1 --warmup 1 $FLAGS --precision $P (full-HIP AV step, batch 64; one stream)" > gpurun_out/ck/${TAG}_sq_counters_$P.txt
  AVSEP_FORK_SOURCES=1 AVSEP_FORK_PAIR=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --precision $P --layers gpurun_out/ck/${TAG}_layers_$P.txt > gpurun_out/ck/layers_$P.log 2>&1
  rm -rf gpurun_out/ck/tr_$P gpurun_out/ck/f_$P gpurun_out/ck/w_$P gpurun_out/ck/sq_$P
  echo done $P
done
# configs[4] (3 sources, 5 frames, 512x256, batch 32): the PMC traffic passes of its headline precision
P=f32
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/ck/f5 -- python3 bench.py --config 5 --steps 2 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/f5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/ck/w5 -- python3 bench.py --config 5 --steps 2 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/w5.log 2>&1
python profiles/summarise_pmc.py gpurun_out/ck/f5 gpurun_out/ck/w5 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --config 5 --steps 2 --warmup 1 $FLAGS --precision $P" 3 32 gpurun_out/ck/pmc_${P}_config5.json > gpurun_out/ck/pmc_config5.log
rm -rf gpurun_out/ck/f5 gpurun_out/ck/w5
echo done config5
# the step as shipped (independent passes on forked streams): a kernel trace for the record — per-kernel durations are NOT exclusive here
export AVSEP_FORK_SOURCES=1 AVSEP_FORK_PAIR=1
for P in f32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ck/trf_$P -- python3 bench.py --steps 3 --warmup 1 $FLAGS --precision $P > gpurun_out/ck/trf_$P.log 2>&1
  python profiles/summarise_trace.py gpurun_out/ck/trf_$P "python3 bench.py --steps 3 --warmup 1 $FLAGS --precision $P (the step as shipped: passes on forked streams; kernels of different streams overlap, durations are not exclusive)" > gpurun_out/ck/${TAG}_bench_kernel_stats_${P}_forked.txt
  rm -rf gpurun_out/ck/trf_$P
done
echo done forked
