"""cfg5 per-GPU shard, aggregation only (for rocprofv3 --kernel-trace --stats: main kernel vs hub kernel time)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import ops, workloads  # noqa: E402

N, E, M = 1_250_000, 12_500_000, 256
adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
adj = [(adj[0][0].cuda(), adj[0][1].cuda())]
y = torch.randn(N, M, device="cuda")
plan = ops.build_plan(adj, N)
for red in ("sum", "max"):
    for _ in range(6):
        ops.gather_reduce(y, plan, M, red)
torch.cuda.synchronize()
print("done")
