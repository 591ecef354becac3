#!/bin/bash
# A/B of probe builds on the cfg3 edge GEMM / GRU shapes (run ON THE GPU BOX): scripts/gpu_probe_ab.sh <tag> "<lib tags>" [pytest -k expr]
tag=${1:-ab}; libs=${2:-}; kexpr=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
out=$ROOT/gpurun_out/$tag; mkdir -p $out
cd $ROOT
if [ -n "$kexpr" ]; then
  for lib in $libs; do
    PTGNN_AMD_LIB=$ROOT/ptgnn_amd/csrc/libptgnn_amd_$lib.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_stream.py tests/test_gpu_golden_wide.py -x -q -k "$kexpr" 2>&1 | tail -3 | tee -a $out/tests.log
  done
fi
for round in 1 2; do
for lib in default $libs; do
  if [ "$lib" != default ]; then export PTGNN_AMD_LIB=$ROOT/ptgnn_amd/csrc/libptgnn_amd_$lib.so; else unset PTGNN_AMD_LIB; fi
  for K in 128 64; do
    echo "$lib K=$K $(PROBE_K=$K timeout 120 python scripts/experiments/edge_probe.py 2>&1 | tail -1)" | tee -a $out/probe.log
  done
done
done
