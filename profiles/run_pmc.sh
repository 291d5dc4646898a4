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
# second SQ pass (own run: an unknown counter name must not take the first pass down with it)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_$TAG -o sq2 -- $CMD > /dev/null 2> $R/gpurun_out/pmc_$TAG/sq2.err
cd $R
DBS=""
for f in sq sq2 fetch write; do [ -f gpurun_out/pmc_$TAG/${f}_results.db ] && DBS="$DBS gpurun_out/pmc_$TAG/${f}_results.db"; done
python profiles/summarize_pmc.py gpurun_out/pmc_${TAG}.md $DBS; tail -3 gpurun_out/pmc_$TAG/sq2.err; rm -rf gpurun_out/pmc_$TAG
