"""Weight-gradient GEMM timings (HIP events, median) at the Graph2Class training shapes.  PTGNN_AMD_LIB selects an
A/B build (scripts/build_variant.sh <tag> edge_wgrad.hip -DPTGNN_WGRAD_STEP=..)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops, workloads  # noqa: E402


def t_med(fn, reps=15):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


g = torch.Generator().manual_seed(3)
n = 115772
x = torch.randn(n, 128, generator=g).cuda()
gy = torch.randn(n, 384, generator=g).cuda()
mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
N = mb["num_nodes"]
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]
adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device="cuda")
adj.append((ar, ar))
E = sum(int(a[0].shape[0]) for a in adj)
xe = torch.randn(N, 128, generator=g).cuda()
gm = torch.randn(E, 128, generator=g).cuda()
res = {"lib": os.path.basename(os.environ.get("PTGNN_AMD_LIB", "default"))}
ms = t_med(lambda: ops.linear_weight_grad(x, gy, want_bias=True))
res["dense_116k_384x128"] = {"us": round(ms * 1e3, 1), "frac": round(2.0 * n * 384 * 128 / ms / 1e9 / 157.3, 3)}
ms = t_med(lambda: ops.edge_weight_grad(xe, adj, gm, False))
res["edge_cfg3_T17_128x128"] = {"us": round(ms * 1e3, 1), "frac": round(2.0 * E * 128 * 128 / ms / 1e9 / 157.3, 3)}
ms = t_med(lambda: ops.edge_weight_grad(xe, adj, gm, False, dropout_p=0.1, dropout_seed=5))
res["edge_cfg3_dropout"] = {"us": round(ms * 1e3, 1), "frac": round(2.0 * E * 128 * 128 / ms / 1e9 / 157.3, 3)}
print(json.dumps(res))
