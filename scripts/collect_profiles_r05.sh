#!/bin/bash
# Run in the authoring container after `gpurun -- bash scripts/gpu_r05_final.sh` merged its output: copies the
# summaries the notes / bench.py cite from gpurun_out/ (scratch) into profiles/ (tracked).
cd "$(dirname "$0")/.."
for t in cfg2 cfg3 cfg4 cfg5 train_ggnn train_mlp; do
  src=gpurun_out/prof_r05_$t
  [ -d $src ] || { echo "missing $src"; continue; }
  cp $src/summary.md profiles/r05_${t}_rocprofv3_summary.md
  f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/r05_${t}_kernel_stats.csv
  [ -f $src/traffic.json ] && cp $src/traffic.json profiles/r05_${t}_traffic.json
done
[ -f gpurun_out/pmc_r05_cfg3_duty/summary.txt ] && cp gpurun_out/pmc_r05_cfg3_duty/summary.txt profiles/r05_cfg3_pmc_duty.txt
python - "$1" <<'PY'
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05final"
d = json.loads([l for l in open(f"gpurun_out/{tag}/bench.json") if l.startswith("{")][-1])
json.dump(d, open("profiles/r05_bench_n1.json", "w"), indent=1)
PY
ls -la profiles | grep r05
