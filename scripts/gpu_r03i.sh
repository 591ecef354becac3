#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
timeout 600 python -m pytest tests -x -q -m gpu -k "weight_grad or training_gradients or long_rows or hub or capturable" > $out/focus.log 2>&1; tail -4 $out/focus.log
for v in "" _s32 _s64; do PTGNN_AMD_LIB=$PWD/ptgnn_amd/csrc/libptgnn_amd$v.so python scripts/wgrad_bench.py 2>&1 | tail -1 >> $out/wgrad.log; done
cat $out/wgrad.log
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; grep -i "train\|primary" $out/bench.err
