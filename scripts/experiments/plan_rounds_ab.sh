#!/bin/bash
# A/B of the split kernels' sub-tile size (records per workgroup = 512 threads x PTGNN_SPLIT_ROUNDS) on every BASELINE plan shape
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_k
for lib in "" ptgnn_amd/csrc/libptgnn_amd_rounds4.so ptgnn_amd/csrc/libptgnn_amd_rounds2.so; do
  PTGNN_AMD_LIB=$lib python scripts/plan_bench.py --reps 40 2>/dev/null | tail -1
done > gpurun_out/r06_k/plan.log 2>&1
cat gpurun_out/r06_k/plan.log
