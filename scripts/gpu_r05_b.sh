#!/bin/bash
# Round 5, call B: row-range aggregation + aggregation/GRU pipeline -- parity tests, then the headline step A/B over the
# number of row ranges (1 = the unsplit pair of round 4).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullrow.py -x -q -m gpu -k "pipeline or row_range or mlp-sum or pipelined" > $out/tests.log 2>&1
echo "tests rc=$?" >> $out/tests.log; tail -4 $out/tests.log
for p in 1 2 3 4 1 2; do
  PTGNN_AMD_AGG_PIPELINE=$p timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --no-sustained > $out/bench_p$p.json 2> $out/bench_p$p.err
  python - $out/bench_p$p.json $p <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("pieces", sys.argv[2], "ms_per_step", d["ms_per_step"], "repeats", d.get("repeats", {}).get("ms_per_step_median"))
PY
done
