#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace stats of the Graph2Class-shaped training step.
# Usage: scripts/train_profile.sh <tag> <dropout>
set -u
TAG=${1:-train}; DROP=${2:-0.1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TRAIN_DROPOUTS=$DROP TRAIN_ARCH=${TRAIN_ARCH:-ggnn} timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/scripts/train_bench.py > "$OUT/trace.log" 2>&1
python "$ROOT/scripts/summarize_prof.py" "$OUT" > "$OUT/summary.md" 2>&1
tail -2 "$OUT/trace.log"
head -40 "$OUT/summary.md"
