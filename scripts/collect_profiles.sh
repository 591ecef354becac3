#!/bin/bash
# Run in the authoring container after `gpurun -- bash scripts/gpu_profile_r06.sh` merged its output: copies the summaries the
# notes / bench.py cite from gpurun_out/ (scratch) into profiles/ (tracked).  usage: scripts/collect_profiles.sh <round tag, e.g. r06> [bench dir under gpurun_out]
cd "$(dirname "$0")/.."
R=${1:-r06}
for t in cfg1 cfg2 cfg3 cfg4 cfg4_ggnn cfg5 train_ggnn train_mlp; do
  src=gpurun_out/prof_${R}_$t
  [ -d $src ] || { echo "missing $src"; continue; }
  cp $src/summary.md profiles/${R}_${t}_rocprofv3_summary.md
  f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_${t}_kernel_stats.csv
  [ -f $src/traffic.json ] && cp $src/traffic.json profiles/${R}_${t}_traffic.json
done
if [ -n "${2:-}" ]; then
python - "$R" "$2" <<'PY'
import json, sys
rnd, tag = sys.argv[1], sys.argv[2]
d = json.loads([l for l in open(f"gpurun_out/{tag}/bench.json") if l.startswith("{")][-1])
json.dump(d, open(f"profiles/{rnd}_bench_n1.json", "w"), indent=1)
PY
fi
ls -la profiles | grep "$R"
