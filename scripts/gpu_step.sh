#!/bin/bash
# usage: scripts/gpu_step.sh <tag> [pytest -k expression]   (runs on the GPU box through gpurun)
# plan / parity tests first (fast feedback), then the whole GPU suite, then the default bench line
tag=${1:-step}; kexpr=${2:-}
out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
if [ -n "$kexpr" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu -k "$kexpr" > $out/focus.log 2>&1
  echo "focus rc=$?" >> $out/focus.log; tail -15 $out/focus.log
fi
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
timeout 600 python bench.py --sharded-variants --no-cpu-baseline --no-secondary > $out/bench_sharded.json 2> $out/bench_sharded.err
echo "bench sharded rc=$?"; tail -3 $out/bench_sharded.err
