#!/bin/bash
# Round 5, closing call: the whole GPU suite on the final tree + FETCH / WRITE passes for the cfg4 forward (k_gather_update).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05final3; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05_cfg4; mkdir -p $OUT
CMD="python $ROOT/scripts/profile_cfg4.py 20"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
head -40 $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
