#!/bin/bash
# Round 5, last call: the whole GPU suite on the final tree, the training-step profiles again (the K = 3 H GEMMs moved
# off the ring after the first collection), and the default bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05final2; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -6 $out/pytest.log
TRAIN_ARCH=ggnn scripts/train_profile.sh r05_train_ggnn 0.1 > /dev/null 2>&1
TRAIN_ARCH=mlp scripts/train_profile.sh r05_train_mlp 0.1 > /dev/null 2>&1
for t in r05_train_ggnn r05_train_mlp; do echo "== $t"; head -14 gpurun_out/prof_$t/summary.md; done
find gpurun_out/prof_r05_train* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r05_train* -name "*.db" -delete
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
