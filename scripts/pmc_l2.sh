#!/bin/bash
# Run ON THE GPU BOX: L2 (TCC) hit / miss counters per kernel for a bench workload (own PMC pass).
# Usage: scripts/pmc_l2.sh <tag> [bench args...]
set -u
TAG=${1:-l2}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o l2 -- \
  python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $* > "$OUT/pmc_l2.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
f = glob.glob(os.path.join(root, "pmc_l2", "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter file; log tail:"); print(open(os.path.join(root, "pmc_l2.log")).read()[-1500:]); sys.exit(0)
agg = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    import re
    m = re.search(r"(k_\w+)", name)
    if not m:
        continue
    agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
print("| kernel | TCC hits | TCC misses | L2 hit rate |\n|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0))):
    h, m_ = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    if h + m_ > 0:
        print(f"| {k} | {h:.3g} | {m_:.3g} | {100 * h / (h + m_):.1f} % |")
PY
