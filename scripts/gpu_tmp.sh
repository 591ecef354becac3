#!/bin/bash
out=gpurun_out/r03aa; mkdir -p $out
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -x -q -m gpu -k "unique or shared_message or split_gemm or config3" > $out/focus.log 2>&1
echo "focus rc=$?" >> $out/focus.log; grep -E "passed|failed|Error|assert|rc=" $out/focus.log | tail -8
timeout 400 python bench.py --no-secondary > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; grep -E "primary|split" $out/bench.err | head
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03aa/bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["traffic_source"]); s=d.get("split_bf16"); print(s and (s["ms_per_step"], s["parity"], {k:v["avg_ms"] for k,v in s["kernels"].items()}))
PY
