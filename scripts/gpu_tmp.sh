#!/bin/bash
out=gpurun_out/r03aj; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 100 python scripts/experiments/edge_probe.py 2> $out/a.err | tee $out/probe.json
timeout 200 python bench.py --no-cpu-baseline --no-secondary > $out/bench.json 2> $out/bench.err; grep -E "primary" $out/bench.err
timeout 400 python -m pytest tests -x -q -m gpu -k "edge or shared or golden or config3 or unique or sharded" > $out/focus.log 2>&1; grep -E "passed|failed" $out/focus.log | tail -1
