#!/bin/bash
# rocprofv3 kernel trace of one training iteration; the timeline is produced on the GPU box.
# Usage: bash profiles/run_iter_trace.sh <tag> [image|all|feature]
TAG=${1:-rX}
SCOPE=${2:-image}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/itrace_$TAG
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/itrace_$TAG -o t -- python $R/profiles/iteration_breakdown.py $SCOPE > $R/gpurun_out/itrace_${TAG}_bench.json 2> $R/gpurun_out/itrace_$TAG/err.log
cd $R
python profiles/iteration_timeline.py gpurun_out/itrace_$TAG/t_results.db gpurun_out/itrace_${TAG}_timeline.md > /dev/null
python profiles/summarize_rocpd.py gpurun_out/itrace_$TAG/t_results.db gpurun_out/itrace_${TAG}_stats.md "rocprofv3 --kernel-trace --stats -- python profiles/iteration_breakdown.py $SCOPE" > /dev/null
rm -rf gpurun_out/itrace_$TAG
head -120 gpurun_out/itrace_${TAG}_timeline.md
