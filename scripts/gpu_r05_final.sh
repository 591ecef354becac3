#!/bin/bash
# Round 5, final call: whole GPU suite, the default bench line, and every rocprofv3 summary profiles/r05_* is made from.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05final; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -3 $out/bench.err
scripts/gpu_profile.sh r05_cfg3 > /dev/null 2>&1
scripts/gpu_profile.sh r05_cfg2 --workload cfg2 > /dev/null 2>&1
TRAIN_ARCH=ggnn scripts/train_profile.sh r05_train_ggnn 0.1 > /dev/null 2>&1
TRAIN_ARCH=mlp scripts/train_profile.sh r05_train_mlp 0.1 > /dev/null 2>&1
scripts/pmc_cfg3.sh r05_cfg3_duty > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05_cfg4; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/scripts/profile_cfg4.py 20 > $OUT/trace.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cd $ROOT
for t in r05_cfg3 r05_cfg2 r05_cfg4 r05_train_ggnn r05_train_mlp; do echo "== $t"; head -16 gpurun_out/prof_$t/summary.md; done
cat gpurun_out/pmc_r05_cfg3_duty/summary.txt
find gpurun_out/prof_r05_* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r05_* -name "*.db" -delete
find gpurun_out/prof_r05_* -name "*counter_collection.csv" -size +2M -delete
