"""Does splitting the Graph2Class batch by graphs and running the halves on two HIP streams overlap the HBM-bound
aggregation of one half with the MFMA-bound GEMMs of the other?  Times the 8-layer GGNN forward (plan rebuilt every
step, as bench.py does): whole batch / two halves back to back on one stream / two halves on two streams.
Run ON THE GPU BOX."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import layers as L, ops, workloads  # noqa: E402
from ptgnn_amd.gnn import GraphNeuralNetwork  # noqa: E402

dev = torch.device("cuda")
H, T = 128, 17
torch.manual_seed(1234)
ggnn = L.GatedMessagePassingLayer(H, H, T, "max")
r1 = L.ConcatResidualLayer(H)
last = L.GatedMessagePassingLayer(2 * H, H, T, "max")
net = GraphNeuralNetwork([r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last], torch.nn.Identity(), True,
                         True).to(dev).eval()


def batch(graphs, seed):
    mb = workloads.batched_graphs(graphs, 2500, 8, 2.2, seed=seed)
    n = mb["num_nodes"]
    return {"x": workloads.node_states(n, H, seed=seed).to(dev),
            "adj": [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]],
            "n2g": mb["node_to_graph_idx"].to(dev),
            "refs": {k: v.to(dev) for k, v in mb["reference_node_ids"].items()},
            "refg": {k: v.to(dev) for k, v in mb["reference_node_graph_idx"].items()}, "G": mb["num_graphs"]}


def forward(b):
    return net(node_data={"input": b["x"]}, adjacency_lists=b["adj"], edge_feature_data=[], node_to_graph_idx=b["n2g"],
               reference_node_ids=b["refs"], reference_node_graph_idx=b["refg"],
               num_graphs=b["G"]).output_node_representations


def timed(step, n=20, w=5):
    for _ in range(w):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        step()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


whole, a, b = batch(48, 1234), batch(24, 77), batch(24, 78)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
res = {}
with torch.no_grad():
    def step_whole():
        ops.clear_plan_cache()
        forward(whole)

    def step_serial():
        ops.clear_plan_cache()
        forward(a)
        forward(b)

    def step_two_streams():
        ops.clear_plan_cache()
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            forward(a)
        with torch.cuda.stream(s2):
            forward(b)
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    res["whole_48_graphs_ms"] = round(timed(step_whole), 4)
    res["two_halves_one_stream_ms"] = round(timed(step_serial), 4)
    res["two_halves_two_streams_ms"] = round(timed(step_two_streams), 4)
print(json.dumps(res))
