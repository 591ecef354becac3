#!/bin/bash
out=gpurun_out/r03u; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -x -q -m gpu -k "weight_grad or wgrad or training or golden or backward" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; grep -E "passed|failed" $out/focus.log | tail -2
timeout 200 python scripts/wgrad_bench.py 2> $out/wgrad.err | tee $out/wgrad.json
