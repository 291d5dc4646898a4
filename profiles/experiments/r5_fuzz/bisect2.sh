#!/bin/bash
one() { (cd $1 && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-iteration-window 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'])"); }
for k in 1 2 3; do
  one scratch/t_a239e6b a239e6b
  TRASE_BENCH_GC=none one . head_none
  TRASE_BENCH_GC=collect one . head_collect
  TRASE_BENCH_GC=disable one . head_disable
done
