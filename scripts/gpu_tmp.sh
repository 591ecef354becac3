#!/bin/bash
out=gpurun_out/r03ab; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 300 python scripts/experiments/two_stream_probe.py 2> $out/probe.err | tee $out/probe.json; tail -3 $out/probe.err
