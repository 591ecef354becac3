#!/bin/bash
# Round 5, call H: 64-column resident slabs against the panel ring for the GRU backward's K = 384 input-gradient GEMMs.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05h; mkdir -p $out
timeout 600 python scripts/linear_ring_bench.py 2>&1 | tee $out/linear.log | head -12
for bn in 0 64 0 64; do
  if [ $bn = 0 ]; then unset PTGNN_AMD_LINEAR_BN; else export PTGNN_AMD_LINEAR_BN=$bn; fi
  timeout 300 python - <<'PY' 2>&1 | tail -1
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.train_cfg3(torch.device("cuda", 0), 0.1, steps=12)
print("BN", os.environ.get("PTGNN_AMD_LINEAR_BN", "default"), "ggnn train step", r["ms_per_train_step"], "linear avg_ms", r["kernels_over_4_steps"]["linear"]["avg_ms"])
PY
done
