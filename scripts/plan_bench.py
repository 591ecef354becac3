"""Plan-build timings (HIP events around ptgnn_amd_csr_build, median of N) on the BASELINE shapes.
    python scripts/plan_bench.py [--reps 30] [--shapes cfg2,cfg3,cfg3_bwd,cfg5]
Select an A/B build of csr_build.hip with PTGNN_AMD_LIB (scripts/build_variant.sh)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops, workloads  # noqa: E402


def shapes(names, dev):
    out = {}
    if "cfg2" in names:
        adj = workloads.random_graph(200_000, 1_100_000, seed=1234)
        out["cfg2"] = ([(s.to(dev), d.to(dev)) for s, d in adj], 200_000, 0)
    if "cfg3" in names or "cfg3_bwd" in names:
        mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
        n = mb["num_nodes"]
        adj = list(mb["adjacency_lists"])
        adj = adj + [(d, s) for s, d in adj]
        ar = torch.arange(n, dtype=torch.int64)
        adj.append((ar, ar))
        cadj = [(s.to(dev).contiguous(), d.to(dev).contiguous()) for s, d in adj]
        if "cfg3" in names:
            out["cfg3"] = (cadj, n, 0)
        if "cfg3_bwd" in names:
            out["cfg3_bwd"] = (cadj, n * len(cadj), 2)
    if "cfg1" in names:       # one PPI-cap minibatch: ~4.4 k nodes, ~120 k edges after reverse + self augmentation, T = 3
        mb = workloads.batched_graphs(2, 2200, 1, 14.0, seed=7)
        n = mb["num_nodes"]
        adj = list(mb["adjacency_lists"])
        adj = adj + [(d, s) for s, d in adj]
        ar = torch.arange(n, dtype=torch.int64)
        adj.append((ar, ar))
        out["cfg1"] = ([(s.to(dev).contiguous(), d.to(dev).contiguous()) for s, d in adj], n, 0)
    if "cfg4" in names:
        mb = workloads.batched_graphs(40, 2000, 10, 2.4, seed=21)
        n = mb["num_nodes"]
        adj = list(mb["adjacency_lists"])
        adj = adj + [(d, s) for s, d in adj]
        ar = torch.arange(n, dtype=torch.int64)
        adj.append((ar, ar))
        out["cfg4"] = ([(s.to(dev).contiguous(), d.to(dev).contiguous()) for s, d in adj], n, 0)
    if "cfg5" in names:
        adj = workloads.power_law_graph(1_250_000, 12_500_000, alpha=0.8, seed=1234)
        out["cfg5"] = ([(s.to(dev), d.to(dev)) for s, d in adj], 1_250_000, 0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--shapes", default="cfg1,cfg4,cfg2,cfg3,cfg3_bwd,cfg5")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {"lib": os.environ.get("PTGNN_AMD_LIB", "default")}
    for name, (adj, rows, mode) in shapes(a.shapes.split(","), dev).items():
        for _ in range(3):
            ops.build_plan(adj, rows, mode=mode)
        evs = []
        for _ in range(a.reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.build_plan(adj, rows, mode=mode)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        t = sorted(x.elapsed_time(y) for x, y in evs)
        res[name] = {"median_us": round(t[len(t) // 2] * 1e3, 1), "min_us": round(t[0] * 1e3, 1),
                     "edges": sum(int(x[0].shape[0]) for x in adj), "rows": rows}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
