#!/bin/bash
# Run ON THE GPU BOX (round 5, call A): the new every-row cfg5 parity + facade-install tests, and the rocprofv3
# evidence for the cfg5 shard on the current gather_reduce.hip (kernel-trace stats; FETCH / WRITE in separate passes).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r05a
timeout 1500 python -m pytest tests/test_gpu_fullrow.py tests/test_gpu_facade_install.py -x -q -m gpu -s > gpurun_out/r05a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05a/tests.log
tail -5 gpurun_out/r05a/tests.log
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05_cfg5; mkdir -p $OUT
CMD="python $ROOT/scripts/profile_cfg5.py 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
head -40 $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
cd $ROOT
