#!/usr/bin/env python
"""AUTHORING-CONTAINER ONLY (needs /root/reference): time the reference's OWN modules (with the torch_scatter /
dpu_utils stand-ins of oracle/shims.py) on the bench workloads' bounded CPU samples, and the oracle restatement
beside them.  The numbers go into BASELINE.md section 3; bench.py's `cpu_baseline` (kind "port") is the oracle
because the reference checkout does not exist on the GPU box.
    PYTHONHASHSEED=0 python scripts/ref_cpu_baseline.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shims  # noqa: E402

shims.install()
from ptgnn.neuralmodels.gnn import GraphNeuralNetwork  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing import GatedMessagePassingLayer, MlpMessagePassingLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing.residuallayers import ConcatResidualLayer  # noqa: E402

from oracle import mp_oracle as O  # noqa: E402
from ptgnn_amd import workloads  # noqa: E402


class _Identity(torch.nn.Module):
    def forward(self, input):
        return input


def median_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2]


def main():
    print(f"host cpus = {os.cpu_count()}, torch {torch.__version__}")
    H, T = 128, 17
    mb = workloads.batched_graphs(8, 2500, 8, 2.2, seed=1234)
    n = mb["num_nodes"]
    x = workloads.node_states(n, H, seed=5)
    torch.manual_seed(1234)
    ggnn = GatedMessagePassingLayer(H, H, T, "max")
    r1 = ConcatResidualLayer(H)
    last = GatedMessagePassingLayer(2 * H, H, T, "max")
    mods = [r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last]
    net = GraphNeuralNetwork(mods, _Identity(), True, True).eval()
    specs = ([{"kind": "residual_origin", "name": "r1"}] + [O.weights_from_reference_layer(ggnn)] * 7
             + [{"kind": "residual_concat", "name": "r1"}, O.weights_from_reference_layer(last)])
    e = 2 * sum(int(a[0].shape[0]) for a in mb["adjacency_lists"]) + n

    def ref():
        with torch.no_grad():
            return net(node_data={"input": x}, adjacency_lists=list(mb["adjacency_lists"]), edge_feature_data=[],
                       node_to_graph_idx=mb["node_to_graph_idx"], reference_node_ids={}, reference_node_graph_idx={},
                       num_graphs=mb["num_graphs"]).output_node_representations

    def orc():
        with torch.no_grad():
            return O.gnn_forward(x, mb["adjacency_lists"], specs, True, True)[0]
    print(f"cfg3 sample: 8 graphs, N={n}, E={e}, 8 GGNN layers H=128 max; max |ref - oracle| = "
          f"{float((ref() - orc()).abs().max()):.2e}")
    for threads in sorted({1, min(8, os.cpu_count()), os.cpu_count()}):
        torch.set_num_threads(threads)
        tr, to = median_of(ref), median_of(orc)
        print(f"  threads={threads}: reference modules {tr:.3f} s = {e / (tr / 8) / 1e6:.3f} M edges/s/layer | "
              f"oracle {to:.3f} s = {e / (to / 8) / 1e6:.3f} M edges/s/layer")
    # config 2
    N, E = 200_000, 1_100_000
    adj = workloads.random_graph(N, E)
    x2 = workloads.node_states(N, 128)
    torch.manual_seed(1234)
    layer = MlpMessagePassingLayer(128, 128, 128, 1, "sum").eval()
    spec = O.weights_from_reference_layer(layer)
    feats = [torch.empty(E, 0)]
    for threads in sorted({1, min(8, os.cpu_count()), os.cpu_count()}):
        torch.set_num_threads(threads)
        with torch.no_grad():
            tr = median_of(lambda: layer(x2, adj, None, {}, {}, feats))
            to = median_of(lambda: O.mlp_mp_layer(x2, adj, feats, spec))
        print(f"cfg2 threads={threads}: reference layer {tr:.3f} s = {E / tr / 1e6:.3f} M edges/s | oracle {to:.3f} s = "
              f"{E / to / 1e6:.3f} M edges/s")


if __name__ == "__main__":
    main()
