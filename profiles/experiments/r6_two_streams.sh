#!/bin/bash
# One gpurun call: the round-6 co-residency experiments.  Output: gpurun_out/r6_two_streams.jsonl
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_two_streams.jsonl; mkdir -p gpurun_out; : > $O
V=$PWD/trase_amd/lib/variants
for lib in "" $V/libtrase_rast_noslab.so $V/libtrase_rast_noslp.so $V/libtrase_rast_prio.so; do
  TRASE_RAST_LIB=$lib timeout 240 python profiles/experiments/r6_two_streams.py streams 2>/dev/null | tail -1 >> $O
done
# two processes sharing the GPU: the aggressor loops while the victim runs alone on its own default stream
rm -f /tmp/agg_up
( timeout 200 python profiles/experiments/r6_two_streams.py aggressor 60 2>/dev/null | tail -1 >> $O ) &
sleep 25; touch /tmp/agg_up
timeout 200 python profiles/experiments/r6_two_streams.py victim /tmp/agg_up 2>/dev/null | tail -1 >> $O
wait
# control: the victim process alone
timeout 200 python profiles/experiments/r6_two_streams.py victim 2>/dev/null | tail -1 >> $O
cat $O
