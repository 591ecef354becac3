#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
for v in "" _b8; do PTGNN_AMD_LIB=$PWD/ptgnn_amd/csrc/libptgnn_amd$v.so python scripts/plan_bench.py 2>/dev/null | tail -1 >> $out/plan_bench.log; done; cat $out/plan_bench.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -5 $out/bench.err
