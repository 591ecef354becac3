#!/bin/bash
# Round 5, call L: the dst-range-sharded variants of bench.py over RCCL at world = 1 (what the driver's SCALE run exercises at N > 1).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
out=gpurun_out/r05l; mkdir -p $out
timeout 900 python bench.py --sharded-variants --no-cpu-baseline --no-secondary --no-sustained > $out/bench_sharded.json 2> $out/bench_sharded.err
echo "bench sharded rc=$?"; tail -4 $out/bench_sharded.err
python - $out/bench_sharded.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(json.dumps(d["config"].get("dst_range_split"), indent=0)[:1500])
PY
