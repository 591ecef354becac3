#!/bin/bash
out=gpurun_out/r03ac; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -x -q -m gpu -k "weight_grad or wgrad or training or golden or backward or train" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; grep -E "passed|failed|Error|assert|rc=" $out/focus.log | tail -8
timeout 400 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; grep -E "train|primary" $out/bench.err | head
