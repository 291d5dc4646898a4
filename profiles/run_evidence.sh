#!/bin/bash
# One gpurun call that regenerates the round's evidence at the current sources.  Usage: bash profiles/run_evidence.sh <tag>
# Everything lands in gpurun_out/ (merged back); copy what is to be judged into profiles/.
T=${1:-r6}
R=$PWD
mkdir -p gpurun_out
bash profiles/run_trace.sh $T > /dev/null 2>&1
bash profiles/run_pmc.sh $T sq sq2 fetch write ta > /dev/null 2>&1
python profiles/make_pmc_per_launch.py gpurun_out/pmc_$T.json "profiles/${T}_pmc.md" > /dev/null 2>&1 && cp profiles/pmc_per_launch.json gpurun_out/pmc_per_launch.json
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
python profiles/bench_mlp.py > gpurun_out/${T}_mlp_bench.json 2>/dev/null
python profiles/bench_iteration.py > gpurun_out/${T}_iteration_bench.json 2>/dev/null
python profiles/bench_iteration_feature.py > gpurun_out/${T}_iteration_feature_bench.json 2>/dev/null
python profiles/iteration_breakdown.py image 2>/dev/null | tail -1 > gpurun_out/${T}_iteration_breakdown.json
python profiles/iteration_breakdown.py feature 2>/dev/null | tail -1 >> gpurun_out/${T}_iteration_breakdown.json
bash profiles/run_iter_trace.sh ${T} image > /dev/null 2>&1
bash profiles/run_iter_trace.sh ${T}f feature > /dev/null 2>&1
PMC_CMD="python $R/profiles/bench_mlp.py" PMC_TIMEOUT=200 bash profiles/run_pmc.sh ${T}_mlp fetch write sq2 > /dev/null 2>&1
rm -f gpurun_out/${T}_bench_configs.jsonl
python bench.py --gaussians 1000 --width 128 --height 128 --feat 0 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_bench_configs.jsonl
python bench.py --gaussians 150000 --width 480 --height 270 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_bench_configs.jsonl
python bench.py --gaussians 1000000 --width 1352 --height 1014 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_bench_configs.jsonl
python bench.py --gaussians 2500000 --width 1280 --height 960 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/${T}_bench_configs.jsonl
TRASE_SWEEP_COUNT=${SWEEP:-200} TRASE_SWEEP_SEED=50505 TRASE_FAMILY_COUNT=${FAM:-30} TRASE_FAMILY_SEED=5151 timeout 1500 python -m pytest tests/test_gpu_sweep.py -m gpu -q -s -k "random_parity_sweep or camera_and_scene_families" > gpurun_out/${T}_parity_extended.txt 2>&1
timeout 1300 python -m pytest tests -m gpu -q > gpurun_out/${T}_gputest.txt 2>&1
grep -E "passed|failed" gpurun_out/${T}_gputest.txt gpurun_out/${T}_parity_extended.txt | tail -4
cat gpurun_out/${T}_bench_default.json | cut -c1-400
