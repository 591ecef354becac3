"""Training-step timing of the Graph2Class-style stack on the GPU box (fwd + bwd + Adam)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ptgnn_amd import layers as L, ops, workloads  # noqa: E402
from ptgnn_amd.gnn import GraphNeuralNetwork  # noqa: E402

dev = torch.device("cuda:0")
for dropout in [float(v) for v in os.environ.get("TRAIN_DROPOUTS", "0.0,0.1").split(",")]:
    H, T = 128, 17
    mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
    torch.manual_seed(1234)
    ggnn = L.GatedMessagePassingLayer(H, H, T, "max", dropout_rate=dropout)
    r1 = L.ConcatResidualLayer(H)
    last = L.GatedMessagePassingLayer(2 * H, H, T, "max", dropout_rate=dropout)
    net = GraphNeuralNetwork([r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last], torch.nn.Identity(),
                             True, True).to(dev).train()
    N = mb["num_nodes"]
    E = 2 * sum(int(a[0].shape[0]) for a in mb["adjacency_lists"]) + N
    x = workloads.node_states(N, H, seed=5).to(dev)
    adj = [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]]
    n2g = mb["node_to_graph_idx"].to(dev)
    refs = {k: v.to(dev) for k, v in mb["reference_node_ids"].items()}
    refg = {k: v.to(dev) for k, v in mb["reference_node_graph_idx"].items()}
    head = torch.nn.Linear(2 * H, 100).to(dev)
    opt = torch.optim.Adam(list(net.parameters()) + list(head.parameters()), lr=1e-4)
    target = torch.randint(0, 100, (refs["supernodes"].shape[0],), device=dev)

    def step():
        ops.clear_plan_cache()
        opt.zero_grad(set_to_none=True)
        out = net(node_data={"input": x}, adjacency_lists=adj, edge_feature_data=[], node_to_graph_idx=n2g,
                  reference_node_ids=refs, reference_node_graph_idx=refg, num_graphs=mb["num_graphs"])
        logits = head(out.output_node_representations[out.node_idx_references["supernodes"]])
        loss = torch.nn.functional.cross_entropy(logits, target)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"dropout={dropout}: {dt*1e3:.2f} ms/train step  {E/dt/1e6:.1f} M edges/s (README convention; V100 README training 1.129 M)  "
          f"{mb['num_graphs']/dt:.0f} graphs/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
