#!/bin/bash
tag=${1:-r04h}; libs=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/$tag; mkdir -p $out
cd $ROOT
for round in 1 2; do
for lib in default $libs; do
  if [ "$lib" != default ]; then export PTGNN_AMD_LIB=$ROOT/ptgnn_amd/csrc/libptgnn_amd_$lib.so; else unset PTGNN_AMD_LIB; fi
  timeout 120 python scripts/h64_bench.py 2>&1 | tail -1 | tee -a $out/h64.log
done
done
