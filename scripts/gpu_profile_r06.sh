#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round's rocprofv3 evidence -- kernel-trace stats of the headline command and of
# the secondary workloads, + FETCH_SIZE / WRITE_SIZE passes of the headline (separate --pmc runs).  Summaries land in
# gpurun_out/prof_r06_*/summary.md; scripts/collect_profiles_r06.sh copies them into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
prof() {   # prof <tag> <command...>
  local tag=$1; shift
  local out=$ROOT/gpurun_out/prof_r06_$tag
  mkdir -p "$out"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- "$@" > "$out/trace.log" 2>&1
  python "$ROOT/scripts/summarize_prof.py" "$out" > "$out/summary.md" 2>&1
  cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv" 2>/dev/null
  head -30 "$out/summary.md"
}
BENCH="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-sustained --no-sharded-variants"
bash "$ROOT/scripts/gpu_profile.sh" r06_cfg3 --no-sharded-variants
bash "$ROOT/scripts/gpu_profile.sh" r06_cfg2 --no-sharded-variants --workload cfg2
prof cfg1 python "$ROOT/scripts/profile_cfg1.py" 5
prof cfg4 python "$ROOT/scripts/profile_cfg4.py" 25
prof cfg4_ggnn python "$ROOT/scripts/profile_cfg4.py" 25 ggnn
prof cfg5 python "$ROOT/scripts/profile_cfg5.py"
TRAIN_DROPOUTS=0.0 TRAIN_ARCH=ggnn prof train_ggnn python "$ROOT/scripts/train_bench.py"
TRAIN_DROPOUTS=0.1 TRAIN_ARCH=mlp TRAIN_HIDDEN=64 prof train_mlp python "$ROOT/scripts/train_bench.py"
