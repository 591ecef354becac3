"""Host-side overhead of one TRAINING step (run on the GPU box): issue time vs total, and cProfile of the
step for TRAIN_ARCH=ggnn|mlp."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import graph2class as bench  # noqa: E402
from ptgnn_amd import ops, workloads  # noqa: E402
from ptgnn_amd.gnn import GraphNeuralNetwork  # noqa: E402

arch = os.environ.get("TRAIN_ARCH", "mlp")
H = int(os.environ.get("TRAIN_HIDDEN", "64" if arch == "mlp" else "128"))
dev = torch.device("cuda:0")
mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
torch.manual_seed(1234)
net = GraphNeuralNetwork(bench.typilus_stack(arch, H, 17, 0.1), torch.nn.Identity(), True, True).to(dev).train()
x = workloads.node_states(mb["num_nodes"], H, seed=5).to(dev)
adj = [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]]
n2g = mb["node_to_graph_idx"].to(dev)
refs = {k: v.to(dev) for k, v in mb["reference_node_ids"].items()}
refg = {k: v.to(dev) for k, v in mb["reference_node_graph_idx"].items()}
head = torch.nn.Linear(net.output_node_state_dim, 100).to(dev)
opt = torch.optim.Adam(list(net.parameters()) + list(head.parameters()), lr=1e-4)
target = torch.randint(0, 100, (refs["supernodes"].shape[0],), device=dev)


def step():
    ops.clear_plan_cache()
    opt.zero_grad(set_to_none=True)
    out = net(node_data={"input": x}, adjacency_lists=adj, edge_feature_data=[], node_to_graph_idx=n2g,
              reference_node_ids=refs, reference_node_graph_idx=refg, num_graphs=mb["num_graphs"])
    logits = head(out.output_node_representations[out.node_idx_references["supernodes"]])
    torch.nn.functional.cross_entropy(logits, target).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{arch} H={H}: host issue {(t1 - t0) / 10 * 1e3:.2f} ms/step, total {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
