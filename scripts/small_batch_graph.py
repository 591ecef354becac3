"""Launch-bound regime: a PPI-sized minibatch (~3k nodes, the reference's `ppi/train.py:70` batch cap) through
the shipped PPI architecture shape (5 MLP-MP layers, hidden 256) -- eager launches vs one HIP graph replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import layers as L, ops, workloads  # noqa: E402
from ptgnn_amd.gnn import GraphNeuralNetwork  # noqa: E402

mb = workloads.batched_graphs(2, 1500, 1, 14.0, seed=3)
N, H, T = mb["num_nodes"], 256, 3
torch.manual_seed(0)
net = GraphNeuralNetwork([L.MlpMessagePassingLayer(H, H, H, T, "max") for _ in range(5)], torch.nn.Identity(),
                         True, True).cuda().eval()
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]
n2g = mb["node_to_graph_idx"].cuda()
x = workloads.node_states(N, H, seed=1).cuda()
E = 2 * sum(int(a[0].shape[0]) for a in adj) + N


def fwd():
    ops.clear_plan_cache()
    return net(node_data={"input": x}, adjacency_lists=adj, edge_feature_data=[], node_to_graph_idx=n2g,
               reference_node_ids={}, reference_node_graph_idx={}, num_graphs=mb["num_graphs"]
               ).output_node_representations


def clock(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    t_eager = clock(fwd)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fwd()
    t_graph = clock(g.replay)
print(f"PPI-sized minibatch N={N} E={E} 5xMLP-MP H=256: eager {t_eager*1e3:.3f} ms, HIP-graph replay "
      f"{t_graph*1e3:.3f} ms ({t_eager/t_graph:.2f}x); {E*5/t_graph/1e9:.2f} G edge-messages/s")
