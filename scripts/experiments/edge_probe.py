"""Time the shared-row edge GEMM and the fused GRU on the cfg3 batch (HIP events, median); PTGNN_AMD_LIB selects a probe
build (scripts/build_variant.sh nostore stream_gemm.hip -DPTGNN_PROBE_NOSTORE).  Run ON THE GPU BOX."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import ops, workloads  # noqa: E402


def t_med(fn, reps=21):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
N = mb["num_nodes"]
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]
adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device="cuda")
adj.append((ar, ar))
g = torch.Generator().manual_seed(3)
x = torch.randn(N, 128, generator=g).cuda()
ws = [(torch.randn(128, 128, generator=g) / 11.3).cuda() for _ in adj]
plan = ops.plan_for(adj, N)
uq = plan.unique_messages()
res = {"lib": os.path.basename(os.environ.get("PTGNN_AMD_LIB", "default")), "rows": uq.rows(wait=True)}
res["edge_linear_shared_us"] = round(t_med(lambda: ops.edge_linear_shared(x, uq, ws)) * 1e3, 1)
res["edge_linear_per_edge_us"] = round(t_med(lambda: ops.edge_linear(x, adj, ws, False)) * 1e3, 1)
cell = torch.nn.GRUCell(128, 128).cuda()
a = torch.randn(N, 128, generator=g).cuda()
res["gru_us"] = round(t_med(lambda: ops.gru_cell(a, x, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)) * 1e3, 1)
print(json.dumps(res))
