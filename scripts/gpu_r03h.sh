#!/bin/bash
out=gpurun_out/r03h; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu -k "long_rows or hub or capturable or config5 or weight_grad or training_gradients_match_reference" > $out/focus.log 2>&1; tail -5 $out/focus.log
PTGNN_AMD_HUB_STREAM=0 python scripts/cfg5_gather_bench.py 2>/dev/null | tail -1 >> $out/gather.log
python scripts/cfg5_gather_bench.py 2>/dev/null | tail -1 >> $out/gather.log
cat $out/gather.log
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; grep -i "train\|primary" $out/bench.err
