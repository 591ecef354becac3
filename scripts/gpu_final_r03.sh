#!/bin/bash
# Run ON THE GPU BOX: the round's closing sequence -- GPU suite, smoke, the default bench line, the world-1 sharded
# variants, training profiles of the final tree.
cd "$(dirname "$0")/.."
out=gpurun_out/r03final; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --sharded-variants --no-cpu-baseline --no-secondary > $out/bench_sharded.json 2> $out/bench_sharded.err; echo "bench sharded rc=$?"
TRAIN_ARCH=ggnn scripts/train_profile.sh r03_train_ggnn 0.1 > /dev/null 2>&1
TRAIN_ARCH=mlp scripts/train_profile.sh r03_train_mlp 0.1 > /dev/null 2>&1
find gpurun_out/prof_r03_train_* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r03_train_* -name "*.db" -delete
head -12 gpurun_out/prof_r03_train_mlp/summary.md
