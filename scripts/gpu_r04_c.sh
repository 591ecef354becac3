#!/bin/bash
# usage: gpu_r04_c.sh <tag> "<lib tags>"   -- edge_probe.py for the default build and each probe build, two rounds
tag=${1:-r04c}; libs=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/$tag; mkdir -p $out
cd $ROOT
for round in 1 2; do
for lib in default $libs; do
  if [ "$lib" != default ]; then export PTGNN_AMD_LIB=$ROOT/ptgnn_amd/csrc/libptgnn_amd_$lib.so; else unset PTGNN_AMD_LIB; fi
  timeout 120 python scripts/experiments/edge_probe.py 2>&1 | tail -1 | tee -a $out/probe.log
done
done
