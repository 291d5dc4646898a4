#!/bin/bash
# Second round-6 call: the TWO-PROCESS arrangement (the strong reproducer: 44 of 48 views) against victim and aggressor variants.
# Output: gpurun_out/r6_two_processes.jsonl
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_two_processes.jsonl; mkdir -p gpurun_out; : > $O
V=$PWD/trase_amd/lib/variants
pair() {   # $1 = victim library ("" = product), $2 = aggressor variant bits ("" = product kernels), $3 = label
  rm -f /tmp/agg_up
  ( TRASE_AGG_VARIANT=$2 timeout 120 python profiles/experiments/r6_two_streams.py aggressor 40 > /dev/null 2>&1 ) &
  sleep 22; touch /tmp/agg_up
  echo -n "{\"case\": \"$3\", \"result\": " >> $O
  TRASE_RAST_LIB=$1 timeout 100 python profiles/experiments/r6_two_streams.py victim /tmp/agg_up 2>/dev/null | tail -1 | tr -d '\n' >> $O
  echo "}" >> $O
  wait
}
pair "" "" "product victim, product aggressor"
pair "$V/libtrase_rast_prio.so" "" "victim with s_setprio 3"
pair "$V/libtrase_rast_noslp.so" "" "victim without v_pk_*_f32 (-fno-slp-vectorize)"
pair "$V/libtrase_rast_noslab.so" "" "victim without LDS"
pair "" "0x2040" "product victim, aggressor on the packed-FP32 compositing kernels (no MFMA / transposing reads)"
pair "" "" "product victim, product aggressor (repeat)"
cat $O
