#!/bin/bash
out=gpurun_out/r03z; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -x -q -m gpu -k "unique or shared_message or sharded or two_block or overlap" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; grep -E "passed|failed|Error|assert|rc=" $out/focus.log | tail -8
