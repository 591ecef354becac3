#!/bin/bash
out=gpurun_out/r03b; mkdir -p $out
python scripts/cfg5_parity_diag.py > $out/diag.log 2>&1
for v in "" _w8r4 _w16r4 _w16r2 _w8r8b8 _w16r4b8; do
  PTGNN_AMD_LIB=$PWD/ptgnn_amd/csrc/libptgnn_amd$v.so python scripts/plan_bench.py >> $out/plan_bench.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in "" _w16r4; do
PTGNN_AMD_LIB=/root/repo/ptgnn_amd/csrc/libptgnn_amd$v.so rocprofv3 --kernel-trace --stats -d /root/repo/$out/prof$v -o plan -- python /root/repo/scripts/plan_bench.py --reps 20 > /root/repo/$out/prof$v.log 2>&1
done
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu --deselect "tests/test_gpu_parity.py::test_config5_scaled_layer_vs_oracle" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
cat $out/plan_bench.log
