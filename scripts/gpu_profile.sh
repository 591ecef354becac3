#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: scripts/gpu_profile.sh <tag> [bench args...]   (bench default workload = cfg3; pass --workload cfg2 for config 2)
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-sustained $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
python "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
