#!/bin/bash
# PMC counters for one linear shape (run on the GPU box): scripts/pmc_linear.sh <tag>
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one_linear.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from ptgnn_amd import ops
rows, k, n = 200000, 128, 256
x = torch.randn(rows, k, device="cuda"); w = torch.randn(n, k, device="cuda"); out = torch.empty(rows, n, device="cuda")
for _ in range(6): ops.linear(x, w, out=out)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/a -o a -- python /tmp/one_linear.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o b -- python /tmp/one_linear.py > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in "ab":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print("no csv for", tag); print(open("$OUT/%s.log" % tag).read()[-1500:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_linear" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:34s} last={v[-1]:.4g}  n={len(v)}")
PY
