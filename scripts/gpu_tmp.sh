#!/bin/bash
out=gpurun_out/r03v; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -x -q -m gpu -k "linear_ring or streaming_weight_grad or linear" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; grep -E "passed|failed|Error|assert" $out/focus.log | tail -6
timeout 300 python scripts/linear_ring_bench.py 2> $out/ring.err | tee $out/ring.json
