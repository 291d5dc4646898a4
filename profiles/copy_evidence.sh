#!/bin/bash
# After `gpurun -- bash profiles/run_evidence.sh <tag>`: copy what is to be judged from gpurun_out/ into profiles/.
T=${1:-r6}
cd "$(dirname "$0")/../gpurun_out" || exit 1
cp trace_$T.md ../profiles/${T}_kernel_stats.md
cp trace_${T}_bench.json ../profiles/${T}_bench_under_rocprof.json
cp pmc_$T.md ../profiles/${T}_pmc.md
cp pmc_per_launch.json ../profiles/pmc_per_launch.json
cp pmc_${T}_mlp.md ../profiles/${T}_pmc_mlp.md
for f in bench_default.json mlp_bench.json iteration_bench.json iteration_feature_bench.json iteration_breakdown.json bench_configs.jsonl parity_extended.txt; do
  cp ${T}_$f ../profiles/${T}_$f
done
cp itrace_${T}_timeline.md ../profiles/${T}_iteration_timeline.md
cp itrace_${T}f_timeline.md 2>/dev/null ||:; true # ../profiles/${T}_iteration_feature_timeline.md
grep -E "passed|failed" ${T}_gputest.txt > ../profiles/${T}_gputest.txt
cd .. && python - <<'PY'
import json
from bench import source_sha16
t = json.load(open("profiles/pmc_per_launch.json"))
print("pmc table sources", t.get("_source_sha16"), "tree", source_sha16(), "MATCH" if t.get("_source_sha16") == source_sha16() else "MISMATCH")
PY
