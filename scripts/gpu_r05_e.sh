#!/bin/bash
# Round 5, call E: the three forms of the fused aggregation + node update (bit-identity tests, cfg4 A/B).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gather_update.py -x -q -m gpu > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -6 $out/tests.log
for f in off 4 16 32 off 4 16; do
  if [ $f = off ]; then export PTGNN_AMD_GATHER_UPDATE=0; else export PTGNN_AMD_GATHER_UPDATE=1 PTGNN_AMD_GATHER_UPDATE_MFMA=$f; fi
  timeout 300 python scripts/profile_cfg4.py 40 2>&1 | tail -1 | sed "s/^.*unsharded:/form=$f/"
done
export PTGNN_AMD_GATHER_UPDATE=1 PTGNN_AMD_GATHER_UPDATE_MFMA=4
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05_cfg4; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/scripts/profile_cfg4.py 20 > $OUT/trace.log 2>&1
PROF_TOP=25 python $ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.md 2>&1
head -14 $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
