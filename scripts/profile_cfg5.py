"""BASELINE config 5 per-GPU shard (power-law N=1.25M, E=12.5M, H=M=256): one GGNN and one MLP-MP layer step
(plan build + layer), the loop rocprofv3 wraps for profiles/r03_cfg5_*.  usage: python scripts/profile_cfg5.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import layers as L, ops, workloads  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, E, H = 1_250_000, 12_500_000, 256
adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
cadj = [(adj[0][0].cuda(), adj[0][1].cuda())]
x = workloads.node_states(N, H, seed=2).cuda()
torch.manual_seed(5)
for layer in (L.GatedMessagePassingLayer(H, H, 1, "sum"), L.MlpMessagePassingLayer(H, H, H, 1, "sum")):
    layer = layer.cuda().eval()
    for _ in range(iters):
        ops.clear_plan_cache()
        with torch.no_grad():
            layer(x, cadj, None, {}, {}, [None])
torch.cuda.synchronize()
print("done")
