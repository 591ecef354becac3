#!/bin/bash
# usage: scripts/gpu_step.sh <tag> [pytest -k expression] [skip-suite]   (runs on the GPU box through gpurun)
# focus tests first (fast feedback), then the whole GPU suite, then the default bench line
tag=${1:-step}; kexpr=${2:-}; skip=${3:-}
out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
if [ -n "$kexpr" ]; then
  timeout 1200 python -m pytest tests -x -q -m gpu -k "$kexpr" -s > $out/focus.log 2>&1
  echo "focus rc=$?" >> $out/focus.log; tail -25 $out/focus.log
fi
if [ -z "$skip" ]; then
  timeout 1800 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1
  echo "pytest rc=$?" >> $out/pytest.log; tail -15 $out/pytest.log
fi
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -5 $out/bench.err
