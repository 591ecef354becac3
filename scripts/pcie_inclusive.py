"""PCIe-inclusive rate: the Graph2Class-sized batch handed over as HOST buffers --
per-graph int32 edge arrays through MinibatchBuilder.finalize (one pinned staging upload + one launch) and the
node states from pinned host memory -- then the 8-layer forward."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import graph2class as bench  # noqa: E402
from ptgnn_amd import ops  # noqa: E402
from ptgnn_amd.batching import MinibatchBuilder  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.RandomState(1234)
T0, graphs = 8, []
for _ in range(48):
    n = int(rng.randint(1900, 3000))
    adj = []
    for t in range(T0):
        e = int(2.2 * n * (0.35 if t == 0 else 0.65 / 7))
        adj.append((rng.randint(0, n, e).astype(np.int32), rng.randint(0, n, e).astype(np.int32)))
    graphs.append((adj, n, {"supernodes": rng.randint(0, n, 20).astype(np.int32)}))
N = sum(g[1] for g in graphs)
E_raw = sum(a[0].shape[0] for g in graphs for a in g[0])
E = 2 * E_raw + N
H = 128
x_host = torch.randn(N, H).pin_memory()
torch.manual_seed(0)
from ptgnn_amd.gnn import GraphNeuralNetwork  # noqa: E402
net = GraphNeuralNetwork(bench.typilus_stack("ggnn", H, 17, 0.0), torch.nn.Identity(), True, True).to(dev).eval()


def handover():
    b = MinibatchBuilder(T0, 10 ** 9)
    for adj, n, refs in graphs:
        b.extend(adj, n, refs)
    mb = b.finalize(dev)
    return mb, x_host.to(dev, non_blocking=True)


def forward(mb, x):
    ops.clear_plan_cache()
    with torch.no_grad():
        return net(node_data={"input": x}, adjacency_lists=mb["adjacency_lists"], edge_feature_data=[],
                   node_to_graph_idx=mb["node_to_graph_idx"], reference_node_ids=mb["reference_node_ids"],
                   reference_node_graph_idx=mb["reference_node_graph_idx"], num_graphs=mb["num_graphs"])


def clock(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


mb, x = handover()
t_fwd = clock(lambda: forward(mb, x))
t_hand = clock(handover)
t_all = clock(lambda: forward(*handover()))
print(f"N={N} E={E}: resident forward {t_fwd:.3f} ms | host hand-over (pack + {x_host.numel() * 4 / 1e6:.0f} MB states + "
      f"{(2 * E_raw) * 4 / 1e6:.1f} MB int32 indices over PCIe + assembly launch) {t_hand:.3f} ms | hand-over + forward "
      f"{t_all:.3f} ms => {E / (t_all * 1e-3 / 8) / 1e9:.2f} G edges/s per layer PCIe-inclusive vs {E / (t_fwd * 1e-3 / 8) / 1e9:.2f} G resident")
