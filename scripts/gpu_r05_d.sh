#!/bin/bash
# Round 5, call D: fused aggregation + node update (hidden 64) -- parity tests, cfg4 / README-architecture A/B.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05d; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gather_update.py tests/test_gpu_pipeline.py tests/test_gpu_two_rank.py -x -q -m gpu > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -6 $out/tests.log
for f in 0 1 0 1 32; do
  PTGNN_AMD_GATHER_UPDATE_MFMA=$f PTGNN_AMD_GATHER_UPDATE=$f timeout 300 python scripts/profile_cfg4.py 40 2>&1 | tail -1 | sed "s/^/fused=$f /"
done
for f in 0 1; do
  PTGNN_AMD_GATHER_UPDATE=$f timeout 300 python - <<'PY' 2>&1 | tail -2
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
r = bench.train_cfg3(torch.device("cuda", 0), 0.1, arch="mlp", H=64, forward_too=True)
print("fused", os.environ["PTGNN_AMD_GATHER_UPDATE"], "readme arch train", r["ms_per_train_step"], "forward", r["ms_per_forward"])
k = r["kernels_over_4_steps"]
PY
done
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05_cfg4; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/scripts/profile_cfg4.py 20 > $OUT/trace.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
head -30 $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
