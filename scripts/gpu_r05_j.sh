#!/bin/bash
# Round 5, call J: README-architecture training step three times in fresh processes (was the 9.8 ms of the last bench run a
# hiccup?), then the default bench line of the final tree.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05j; mkdir -p $out
for i in 1 2 3; do
  timeout 300 python - <<'PY' 2>&1 | tail -1
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.train_cfg3(torch.device("cuda", 0), 0.1, arch="mlp", H=64, forward_too=True)
print("readme arch train", r["ms_per_train_step"], r["ms_per_train_step_blocks"], "forward", r["ms_per_forward"])
PY
done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; grep "train" $out/bench.err | tail -4
