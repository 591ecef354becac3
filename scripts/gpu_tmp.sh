#!/bin/bash
out=gpurun_out/r03p; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -x -q -m gpu -k "weight_grad or wgrad or training or golden or backward" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; tail -6 $out/focus.log
timeout 200 python scripts/wgrad_bench.py > $out/wgrad_stream.json 2> $out/wgrad_stream.err; cat $out/wgrad_stream.json
PTGNN_AMD_WGRAD_STREAM=0 timeout 200 python scripts/wgrad_bench.py > $out/wgrad_tile.json 2> $out/wgrad_tile.err; cat $out/wgrad_tile.json
timeout 400 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; grep -i "train" $out/bench.err | tail -5
