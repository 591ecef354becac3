#!/bin/bash
# PMC counters of the dense kernels per GEMM mode (run ON THE GPU BOX): scripts/pmc_probe.sh <tag> [gru|edge|linear]
TAG=${1:-x}; WHAT=${2:-gru}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/probe_one.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from ptgnn_amd import ops, workloads
what = "$WHAT"
g = torch.Generator().manual_seed(3)
dev = "cuda"
if what == "gru":
    n, m, h = 115772, 128, 128
    a = torch.randn(n, m, generator=g).to(dev); hh = torch.randn(n, h, generator=g).to(dev)
    cell = torch.nn.GRUCell(m, h).to(dev)
    f = lambda: ops.gru_cell(a, hh, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)
elif what == "linear":
    x = torch.randn(200000, 128, generator=g).to(dev); w = torch.randn(256, 128, generator=g).to(dev)
    f = lambda: ops.linear(x, w)
else:
    mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
    N = mb["num_nodes"]
    adj = [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]]
    adj = adj + [(d, s) for s, d in adj]
    ar = torch.arange(N, device=dev); adj.append((ar, ar))
    x = torch.randn(N, 128, generator=g).to(dev)
    ws = [(torch.randn(128, 128, generator=g) / 11.3).to(dev) for _ in adj]
    f = lambda: ops.edge_linear(x, adj, ws, False)
for mode in (0, 1, 2):
    ops.set_gemm_mode(mode)
    for _ in range(5): f()
    torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python /tmp/probe_one.py > $OUT/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o b -- python /tmp/probe_one.py > $OUT/b.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python /tmp/probe_one.py > $OUT/t.log 2>&1
python - <<PY
import csv, glob, collections
for tag in "ab":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print("no csv for", tag); print(open("$OUT/%s.log" % tag).read()[-1500:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"]
        if any(s in kn for s in ("k_gru", "k_stream", "k_linear", "k_edge")):
            agg[kn[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, d in agg.items():
        print("==", kn)
        for k, v in d.items():
            print(f"   {k:30s} last={v[-1]:.5g}  n={len(v)}")
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if any(s in r["Name"] for s in ("k_gru", "k_stream", "k_linear", "k_edge")):
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
