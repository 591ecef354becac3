#!/bin/bash
# A/B: two workgroups per CU for the grouped per-edge GEMM at short K (PTGNN_AMD_EDGE_WGS2_MAXK = 0 off | 64 | 128)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_l
for k in 0 64 128; do
  echo "== PTGNN_AMD_EDGE_WGS2_MAXK=$k"
  PTGNN_AMD_EDGE_WGS2_MAXK=$k TRAIN_DROPOUTS=0.1 TRAIN_ARCH=mlp TRAIN_HIDDEN=64 TRAIN_STEPS=10 python scripts/train_bench.py 2>/dev/null | tail -1
  PTGNN_AMD_EDGE_WGS2_MAXK=$k TRAIN_DROPOUTS=0.0 TRAIN_ARCH=ggnn TRAIN_STEPS=8 python scripts/train_bench.py 2>/dev/null | tail -1
  PTGNN_AMD_EDGE_WGS2_MAXK=$k python scripts/profile_cfg4.py 25 mlp 2>/dev/null | tail -1
done > gpurun_out/r06_l/ab.log 2>&1
cat gpurun_out/r06_l/ab.log
