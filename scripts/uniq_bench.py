"""Kernel times of the shared-message-row bookkeeping (ptgnn_amd_unique_sources) on the cfg3 batch.  Run ON THE GPU BOX
(under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops, workloads  # noqa: E402

mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
N = mb["num_nodes"]
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]
adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device="cuda")
adj.append((ar, ar))
for i in range(12):
    ops.clear_plan_cache()
    plan = ops.plan_for(adj, N)
    ops._UNIQ_SKIP[0] = 0
    u = plan.unique_messages()
torch.cuda.synchronize()
print("rows", u.rows(wait=True), "edges", plan.num_edges)
