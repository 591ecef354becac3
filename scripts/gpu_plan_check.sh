#!/bin/bash
out=gpurun_out/r03d; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "csr_build or backward_plan or config5_scaled or hub or plan or capturable" > $out/focus.log 2>&1
tail -4 $out/focus.log
for v in "" _w16r4; do
  PTGNN_AMD_LIB=$PWD/ptgnn_amd/csrc/libptgnn_amd$v.so python scripts/plan_bench.py 2>/dev/null >> $out/plan_bench.log
done
cat $out/plan_bench.log
cd /tmp && export TMPDIR=/tmp
for v in "" _w16r4; do
PTGNN_AMD_LIB=/root/repo/ptgnn_amd/csrc/libptgnn_amd$v.so rocprofv3 --kernel-trace --stats -d /root/repo/$out/prof$v -o plan -- python /root/repo/scripts/plan_bench.py --reps 20 > /root/repo/$out/prof$v.log 2>&1
done
