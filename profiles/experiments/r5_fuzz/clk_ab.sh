#!/bin/bash
run() { python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-iteration-window 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], 'live bwd/fwd', d['roofline']['kernel_ms'], d['roofline']['forward']['kernel_ms'])"; }
for k in 1 2; do
run base
TRASE_BENCH_IDLE_MS=5 run idle5
TRASE_BENCH_IDLE_MS=30 run idle30
TRASE_BENCH_IDLE_MS=200 run idle200
TRASE_BENCH_PREROLL=100 run preroll100
TRASE_BENCH_PREROLL=1000 run preroll1000
STEPS=200 run steps200
STEPS=1000 run steps1000
done
