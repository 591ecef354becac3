#!/bin/bash
# Round 5, call I: 64-column resident slabs adopted -- tests that pin the dispatch, then the training steps.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05i; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_golden_wide.py tests/test_gpu_parity.py -q -m gpu -k "linear or wide or ring or training_gradients or gru or dense" > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -6 $out/tests.log
for bn in 128 0 128 0; do
  if [ $bn = 0 ]; then unset PTGNN_AMD_LINEAR_BN; else export PTGNN_AMD_LINEAR_BN=$bn; fi
  timeout 300 python - <<'PY' 2>&1 | tail -1
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.train_cfg3(torch.device("cuda", 0), 0.1, steps=12)
k = r["kernels_over_4_steps"]["linear"]
print("BN", os.environ.get("PTGNN_AMD_LINEAR_BN", "default(64 where the ring was)"), "ggnn train step", r["ms_per_train_step"], "linear avg_ms", k["avg_ms"], "frac", k["frac"])
PY
done
