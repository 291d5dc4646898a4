#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload in separate passes (gpurun refuses --pmc
# combined with sys/hip traces; kernel-trace is fine).  Usage on the GPU box: bash profiles/run_pmc.sh <tag>
set -u
TAG=${1:-rX}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -d $R/gpurun_out/pmc_$TAG -o sq -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$TAG/sq.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_$TAG -o fetch -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$TAG/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_$TAG -o write -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$TAG/write.err
cd $R
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}.md gpurun_out/pmc_$TAG/sq_results.db gpurun_out/pmc_$TAG/fetch_results.db gpurun_out/pmc_$TAG/write_results.db; rm -rf gpurun_out/pmc_$TAG
