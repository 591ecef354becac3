"""BASELINE configs[0] as the reference batches it (ppi/train.py:66-70: `stop_extending_minibatch_after_num_nodes=3000`,
reverse + self edges => T = 3): 24 PPI-like graphs of 1 500 - 3 500 nodes become ~15-24 minibatches of 3 000 - 6 500 nodes
each -- the SMALL-minibatch regime (tens of dependent launches of a few microseconds), not one 57 k-node batch.

Two stacks over the same minibatches:
  * "ggnn64": what configs[0] names -- ONE GatedMessagePassingLayer(64, 64, 3, "sum");
  * "ppi_arch_mlp256": what ppi/train.py:35-57 ships -- 5 MlpMessagePassingLayer(256) with two mean residuals, sum.
Minibatches are assembled by ptgnn_amd.batching.MinibatchBuilder (device-side finalize) under the reference's stopping
rule; parity replays the same graphs through oracle.mp_oracle.batch_graphs (index tensors bit for bit) and the oracle's
container forward (node states within 1e-5), minibatch by minibatch."""
import time

import torch

from benchmarks.common import PARITY_TOL

NODE_CAP = 3000       # ppi/train.py:70
H_GGNN, H_PPI = 64, 256


def ppi_graphs(seed=1234):
    from ptgnn_amd import workloads
    return workloads.graph_list(24, 1500, 3500, 1, 14.0, seed=seed)


def build_minibatches(graphs, dev):
    """The reference's minibatch loop (abstractneuralmodel.py:290-319) over the device-side builder."""
    from ptgnn_amd.batching import MinibatchBuilder
    out, b = [], MinibatchBuilder(1, NODE_CAP)
    for g in graphs:
        if not b.extend(g["adjacency_lists"], g["num_nodes"], g["reference_nodes"]):
            out.append(b.finalize(dev))
            b = MinibatchBuilder(1, NODE_CAP)
    if len(b):
        out.append(b.finalize(dev))
    return out


def ggnn64_modules(dev):
    from ptgnn_amd import layers as L
    torch.manual_seed(1234)
    layer = L.GatedMessagePassingLayer(H_GGNN, H_GGNN, 3, "sum")
    return [layer.to(dev).eval()], [layer.export_weights()]


def ppi_arch_modules(dev):
    """create_ppi_gnn_model's layer list (ppi/train.py:36-57) at its shipped hidden size 256."""
    from ptgnn_amd import layers as L
    torch.manual_seed(1234)
    mk = lambda: L.MlpMessagePassingLayer(H_PPI, H_PPI, H_PPI, 3, "sum", dropout_rate=0.2)   # noqa: E731
    r1, r2 = L.MeanResidualLayer(H_PPI), L.MeanResidualLayer(H_PPI)
    mods = [r1.pass_through_dummy_layer(), mk(), mk(), mk(), r1, r2.pass_through_dummy_layer(), mk(), mk(), r2]
    marks = [("residual_origin", "r1"), None, None, None, ("residual_mean", "r1"), ("residual_origin", "r2"), None, None,
             ("residual_mean", "r2")]
    specs = [m.export_weights() if mk_ is None else {"kind": mk_[0], "name": mk_[1]} for m, mk_ in zip(mods, marks)]
    return [m.to(dev).eval() for m in mods], specs


def config1(dev, parity=True, passes=5):
    from ptgnn_amd import ops, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    graphs = ppi_graphs()
    mbs = build_minibatches(graphs, dev)
    sizes = [int(mb["node_to_graph_idx"].shape[0]) for mb in mbs]
    edges = [2 * int(mb["adjacency_lists"][0][0].shape[0]) + n for mb, n in zip(mbs, sizes)]
    res = {"workload": f"cfg1: PPI-like, 24 graphs of 1500-3500 nodes (~14 raw links per node, 1 raw edge type -> T=3), batched by "
                       f"the reference's rule (node cap {NODE_CAP}, ppi/train.py:70) into {len(mbs)} minibatches of "
                       f"{min(sizes)}-{max(sizes)} nodes / {min(edges)}-{max(edges)} edges",
           "minibatches": len(mbs), "nodes_per_minibatch_min_max": [min(sizes), max(sizes)],
           "edges_per_minibatch_min_max": [min(edges), max(edges)]}
    ok = True
    for name, builder, hid, n_layers in (("ggnn64", ggnn64_modules, H_GGNN, 1), ("ppi_arch_mlp256", ppi_arch_modules, H_PPI, 5)):
        mods, specs = builder(dev)
        net = GraphNeuralNetwork(mods, torch.nn.Identity(), True, True).to(dev).eval()
        xs_cpu = [workloads.node_states(n, hid, seed=100 + i) for i, n in enumerate(sizes)]
        xs = [x.to(dev) for x in xs_cpu]

        def forward(i):
            mb = mbs[i]
            ops.clear_plan_cache()
            with torch.no_grad():
                return net(node_data={"input": xs[i]}, adjacency_lists=mb["adjacency_lists"], edge_feature_data=[],
                           node_to_graph_idx=mb["node_to_graph_idx"], reference_node_ids=mb["reference_node_ids"],
                           reference_node_graph_idx=mb["reference_node_graph_idx"], num_graphs=mb["num_graphs"])
        for i in range(len(mbs)):
            forward(i)
        blocks = []      # median over the passes (each a sweep over all minibatches)
        for _ in range(passes):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(len(mbs)):
                forward(i)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / len(mbs))
        dt = sorted(blocks)[len(blocks) // 2]
        before = ops.launch_counts()
        timer = ops.KernelTimer()
        ops.set_kernel_timer(timer)
        forward(0)
        ops.set_kernel_timer(None)
        summ = timer.summary()
        launches = sum(d["calls"] for d in summ.values())
        entry = {"mp_layers": n_layers, "hidden": hid, "ms_per_minibatch": round(dt * 1e3, 4),
                 "ms_per_layer": round(dt * 1e3 / n_layers, 4),
                 "edges_per_sec_per_layer": round(sum(edges) / len(edges) / (dt / n_layers), 1),
                 "c_abi_launches_per_minibatch": launches, "c_abi_launches_per_layer": round(launches / n_layers, 2),
                 "device_ms_per_minibatch_sum_of_kernels": round(sum(d["ms"] for d in summ.values()), 4),
                 "kernel_families": sorted(ops.launches_since(before))}
        if parity:
            from oracle import mp_oracle as O
            want_mbs = list(O.batch_graphs(graphs, 1, NODE_CAP))
            assert len(want_mbs) == len(mbs)
            worst, index_ok = 0.0, True
            for i, (mb, wmb) in enumerate(zip(mbs, want_mbs)):
                index_ok = index_ok and torch.equal(mb["node_to_graph_idx"].cpu(), wmb["node_to_graph_idx"]) and all(
                    torch.equal(a.cpu(), b) for pair, wpair in zip(mb["adjacency_lists"], wmb["adjacency_lists"])
                    for a, b in zip(pair, wpair))
                with torch.no_grad():
                    want, n_edges = O.gnn_forward(xs_cpu[i], wmb["adjacency_lists"], specs, True, True)
                got = forward(i).output_node_representations.cpu()
                index_ok = index_ok and n_edges == edges[i]
                worst = max(worst, float((got - want).abs().max()))
            entry["parity"] = {"max_abs": worst, "tol": PARITY_TOL, "strict_1e-5": bool(worst <= PARITY_TOL),
                               "index_tensors_bit_exact": bool(index_ok), "ok": bool(worst <= PARITY_TOL and index_ok),
                               "minibatches_checked": len(mbs),
                               "against": "oracle/mp_oracle.py batch_graphs + gnn_forward, every minibatch"}
            ok = ok and entry["parity"]["ok"]
        res[name] = entry
    res["ok"] = ok
    return res
