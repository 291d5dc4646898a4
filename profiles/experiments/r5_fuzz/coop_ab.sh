#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_sort_coop.py -m gpu -q -x 2>&1 | tail -5
run() { TRASE_RAST_VARIANT=$2 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-iteration-window $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_view']
print('$1', d['value'], d['ms_per_step'], {x:k[x] for x in k if 'radix' in x}, d['launches_per_view'])"; }
for k in 1 2 3; do
run coop 0 ""
run chain 0x800000 ""
done
run coop_S2 0 "--gaussians 150000 --width 480 --height 270 --steps 200 --warmup 20"
run chain_S2 0x800000 "--gaussians 150000 --width 480 --height 270 --steps 200 --warmup 20"
run coop_S2 0 "--gaussians 150000 --width 480 --height 270 --steps 200 --warmup 20"
run chain_S2 0x800000 "--gaussians 150000 --width 480 --height 270 --steps 200 --warmup 20"
