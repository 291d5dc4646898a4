#!/bin/bash
# rocprofv3 kernel trace of the bench command; the summary table is produced on the GPU box because
# the sqlite output is too large to copy back.  Usage: bash profiles/run_trace.sh <tag> [bench args]
TAG=${1:-rX}; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/trace_$TAG
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_$TAG -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/trace_${TAG}_bench.json 2> $R/gpurun_out/trace_$TAG/err.log
cd $R
python profiles/summarize_rocpd.py gpurun_out/trace_$TAG/t_results.db gpurun_out/trace_${TAG}.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline $*" > /dev/null
rm -rf gpurun_out/trace_$TAG
head -40 gpurun_out/trace_${TAG}.md; cat gpurun_out/trace_${TAG}_bench.json | cut -c1-300
