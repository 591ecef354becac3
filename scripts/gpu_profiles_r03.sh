#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 summary profiles/r03_* is made from (kernel-trace stats; FETCH / WRITE PMC passes
# for cfg2 / cfg3 in their own runs), final tree.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
scripts/gpu_profile.sh r03_cfg3 > /dev/null 2>&1
scripts/gpu_profile.sh r03_cfg2 --workload cfg2 > /dev/null 2>&1
TRAIN_ARCH=ggnn scripts/train_profile.sh r03_train_ggnn 0.1 > /dev/null 2>&1
TRAIN_ARCH=mlp scripts/train_profile.sh r03_train_mlp 0.1 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for w in cfg4 cfg5; do
  OUT=$ROOT/gpurun_out/prof_r03_$w; mkdir -p $OUT
  if [ $w = cfg4 ]; then CMD="python $ROOT/scripts/profile_cfg4.py 20"; else CMD="python $ROOT/scripts/profile_cfg5.py 5"; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  if [ $w = cfg5 ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
  fi
  PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
done
cd $ROOT
for t in r03_cfg3 r03_cfg2 r03_cfg4 r03_cfg5 r03_train_ggnn r03_train_mlp; do echo "== $t"; head -24 gpurun_out/prof_$t/summary.md; done
# keep the merge small: drop the raw per-dispatch CSVs / databases, keep stats + summaries + traffic
find gpurun_out/prof_r03_* -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r03_* -name "*.db" -delete
find gpurun_out/prof_r03_* -name "*counter_collection.csv" -size +2M -delete
