"""Where does the scaled-cfg5 layer differ from the fp32 CPU oracle, and is it the kernels or fp32 itself?
Error of the GPU layer and of the fp32 oracle against a float64 oracle evaluation, by in-degree bucket."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mp_oracle as O  # noqa: E402
from ptgnn_amd import layers as L, ops, workloads  # noqa: E402

N, E, H = 125_000, 1_250_000, 256
adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
deg = torch.bincount(adj[0][1], minlength=N)
x = workloads.node_states(N, H, seed=2)
feats = [torch.empty(E, 0)]
edges = [0, 8, 32, 128, 512, 2048, 10 ** 9]
for kind in ("ggnn", "mlp"):
    for agg in ("sum", "max"):
        torch.manual_seed(5)
        layer = (L.GatedMessagePassingLayer(H, H, 1, agg) if kind == "ggnn" else L.MlpMessagePassingLayer(H, H, H, 1, agg)).eval()
        spec = layer.export_weights()
        fn = O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer
        with torch.no_grad():
            w32 = fn(x, adj, feats, spec)
            w64 = fn(x.double(), adj, [f.double() for f in feats], O.cast_spec(spec, torch.float64))
            layer = layer.cuda()
            cadj = [(s.cuda(), d.cuda()) for s, d in adj]
            ops.clear_plan_cache()
            got = layer(x.cuda(), cadj, None, {}, {}, [None]).cpu()
        print(f"== {kind} {agg}: ours-vs-oracle32 {float((got - w32).abs().max()):.3e}  ours-vs-64 "
              f"{float((got.double() - w64).abs().max()):.3e}  oracle32-vs-64 {float((w32.double() - w64).abs().max()):.3e}")
        for lo, hi in zip(edges[:-1], edges[1:]):
            m = (deg >= lo) & (deg < hi)
            if int(m.sum()) == 0:
                continue
            print(f"   deg [{lo},{hi}): rows {int(m.sum()):7d}  ours-vs-32 {float((got - w32)[m].abs().max()):.3e}  "
                  f"ours-vs-64 {float((got.double() - w64)[m].abs().max()):.3e}  oracle32-vs-64 "
                  f"{float((w32.double() - w64)[m].abs().max()):.3e}")
