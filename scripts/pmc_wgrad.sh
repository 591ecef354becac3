#!/bin/bash
# PMC counters of the weight-gradient kernels (run ON THE GPU BOX): scripts/pmc_wgrad.sh <tag>
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/probe_wgrad.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from ptgnn_amd import ops, workloads
g = torch.Generator().manual_seed(3)
dev = "cuda"
n = 115772
x = torch.randn(n, 128, generator=g).to(dev); gy = torch.randn(n, 384, generator=g).to(dev)
mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
N = mb["num_nodes"]
adj = [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]]
adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device=dev); adj.append((ar, ar))
E = sum(int(a[0].shape[0]) for a in adj)
xe = torch.randn(N, 128, generator=g).to(dev); gm = torch.randn(E, 128, generator=g).to(dev)
for _ in range(5):
    ops.linear_weight_grad(x, gy, want_bias=True)
    ops.edge_weight_grad(xe, adj, gm, False)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python /tmp/probe_wgrad.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python /tmp/probe_wgrad.py > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python /tmp/probe_wgrad.py > $OUT/t.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for tag in "ab":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print("no csv for", tag); print(open("$OUT/%s.log" % tag).read()[-1500:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"]
        if "wgrad" in kn:
            agg[kn[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, d in agg.items():
        print("==", kn)
        for k, v in d.items():
            print(f"   {k:30s} last={v[-1]:.5g}  n={len(v)}")
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "wgrad" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
cat $OUT/summary.txt
