#!/bin/bash
# every bench.py mode once (small step counts): exceptions, not numbers
run() { echo "== $*"; timeout 300 python bench.py --steps 4 --warmup 1 --preroll-steps 4 --no-cpu-baseline "$@" 2>gpurun_out/bm.err | tail -1 | cut -c1-160; rc=${PIPESTATUS[0]}; if [ "$rc" != "0" ]; then echo "   rc=$rc"; tail -5 gpurun_out/bm.err; fi; }
run --no-iteration-window
run --policy sync --no-iteration-window
run --graph on --no-iteration-window
run --unfused --no-iteration-window
run --variant 0x400000 --no-iteration-window
run --gaussians 1000 --width 128 --height 128 --feat 0
run --gaussians 150000 --width 480 --height 270
run --shard tiles --no-iteration-window
run --shard tiles --forward-only --no-iteration-window
run --strip-table gpurun_out/strip_tmp.json --no-iteration-window
run --force-collectives --exchange allreduce --no-iteration-window
run --force-collectives --exchange rs_ag --no-iteration-window
run --force-collectives --exchange direct --no-iteration-window
run --force-collectives --exchange direct --exchange-chunks 3 --no-iteration-window
run --force-collectives --exchange phased --active-sh-degree 1 --no-iteration-window
run --force-collectives --bucket accumulate --no-iteration-window
run --force-collectives --bucket sink --no-iteration-window
echo "== torchrun 1 rank"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 1 --preroll-steps 4 --no-cpu-baseline --no-iteration-window 2>gpurun_out/bm.err | tail -1 | cut -c1-160
