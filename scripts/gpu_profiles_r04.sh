#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 summary profiles/r04_* is made from (kernel-trace stats; FETCH / WRITE PMC passes
# for cfg2 / cfg3 in their own runs; PMC duty of the headline step), final tree.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
scripts/gpu_profile.sh r04_cfg3 > /dev/null 2>&1
scripts/gpu_profile.sh r04_cfg2 --workload cfg2 > /dev/null 2>&1
TRAIN_ARCH=ggnn scripts/train_profile.sh r04_train_ggnn 0.1 > /dev/null 2>&1
TRAIN_ARCH=mlp scripts/train_profile.sh r04_train_mlp 0.1 > /dev/null 2>&1
scripts/pmc_cfg3.sh r04_cfg3_duty > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r04_cfg4; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/scripts/profile_cfg4.py 20 > $OUT/trace.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
cd $ROOT
for t in r04_cfg3 r04_cfg2 r04_cfg4 r04_train_ggnn r04_train_mlp; do echo "== $t"; head -24 gpurun_out/prof_$t/summary.md; done
cat gpurun_out/pmc_r04_cfg3_duty/summary.txt
find gpurun_out/prof_r04_* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r04_* -name "*.db" -delete
find gpurun_out/prof_r04_* -name "*counter_collection.csv" -size +2M -delete
