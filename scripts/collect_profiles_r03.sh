#!/bin/bash
# Run in the authoring container after `gpurun -- bash scripts/gpu_profiles_r03.sh` merged its output: copies the
# summaries the notes / bench.py cite from gpurun_out/ (scratch) into profiles/ (tracked).
cd "$(dirname "$0")/.."
for t in cfg2 cfg3 cfg4 cfg5 train_ggnn train_mlp; do
  src=gpurun_out/prof_r03_$t
  [ -d $src ] || { echo "missing $src"; continue; }
  cp $src/summary.md profiles/r03_${t}_rocprofv3_summary.md
  f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r03_${t}_kernel_stats.csv
  [ -f $src/traffic.json ] && cp $src/traffic.json profiles/r03_${t}_traffic.json
done
[ -f gpurun_out/pmc_r03wgrad/summary.txt ] && cp gpurun_out/pmc_r03wgrad/summary.txt profiles/r03_wgrad_pmc.txt
[ -f gpurun_out/pmc_r03wgrad_stream/summary.txt ] && cp gpurun_out/pmc_r03wgrad_stream/summary.txt profiles/r03_wgrad_stream_pmc.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03final/bench.json") if l.startswith("{")][-1])
json.dump(d, open("profiles/r03_bench_n1.json", "w"), indent=1)
s = [l for l in open("gpurun_out/r03final/bench_sharded.json") if l.startswith("{")]
if s:
    json.dump(json.loads(s[-1]), open("profiles/r03_bench_sharded_world1.json", "w"), indent=1)
PY
ls -la profiles | grep r03
