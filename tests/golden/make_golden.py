#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference
(microsoft/ptgnn mounted at /root/reference) executed on CPU in fp32.

Runs only in the authoring container (the reference checkout does not travel to the GPU box).
    PYTHONHASHSEED=0 python tests/golden/make_golden.py

What is the reference's and what is restated:
  * `GatedMessagePassingLayer`, `MlpMessagePassingLayer`, residual layers, `GraphNeuralNetwork`
    and `GraphNeuralNetworkModel` are the reference's own classes, imported unmodified.
  * `torch_scatter` (third-party, absent) is oracle/scatter_ref.py; `dpu_utils` iterators are
    pass-through stubs (oracle/shims.py).
Every fixture stores inputs, the layer weights (oracle spec layout) and the reference output.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import shims  # noqa: E402

shims.install()

from ptgnn.baseneuralmodel import AbstractNeuralModel  # noqa: E402
from ptgnn.neuralmodels.gnn import GraphData, GraphNeuralNetwork, GraphNeuralNetworkModel  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing import (  # noqa: E402
    GatedMessagePassingLayer, MlpMessagePassingLayer, MeanResidualLayer)
from ptgnn.neuralmodels.gnn.messagepassing.residuallayers import ConcatResidualLayer  # noqa: E402
from ptgnn.neuralmodels.gnn.messagepassing import GruGlobalStateUpdate  # noqa: E402
from ptgnn.neuralmodels.reduceops.varsizedsummary import (  # noqa: E402
    SimpleVarSizedElementReduce, WeightedSumVarSizedElementReduce)

from oracle.fixtures import pack_adj, pack_specs  # noqa: E402
from oracle.mp_oracle import weights_from_reference_layer  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand_adj(gen, n, counts):
    adj = []
    for c in counts:
        s = torch.randint(0, n, (c,), generator=gen, dtype=torch.int64)
        d = torch.randint(0, n, (c,), generator=gen, dtype=torch.int64)
        adj.append((s, d))
    return adj


def tricky_adj(gen, n):
    """3 edge types: random with duplicates + a self loop; EMPTY type; a hub destination.
    Nodes n-1 and n-2 never receive an edge (empty segments)."""
    a0 = rand_adj(gen, n - 2, [3 * n])[0]
    a0 = (torch.cat([a0[0], torch.tensor([1, 1, 4])]), torch.cat([a0[1], torch.tensor([2, 2, 4])]))
    a1 = (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64))
    hub_src = torch.randint(0, n, (2 * n,), generator=gen, dtype=torch.int64)
    a2 = (hub_src, torch.full((2 * n,), 3, dtype=torch.int64))
    return [a0, a1, a2]


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v)
                                 for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)} B)")


def empty_feats(adj):
    return [torch.empty(a[0].shape[0], 0) for a in adj]


def single_layers():
    gen = torch.Generator().manual_seed(1234)
    n, H, M = 41, 16, 24
    for agg in ("sum", "mean", "max", "min"):
        torch.manual_seed(100 + len(agg))
        adj = tricky_adj(gen, n)
        x = torch.randn(n, H, generator=gen)
        layer = GatedMessagePassingLayer(H, M, 3, agg).eval()
        with torch.no_grad():
            y = layer(x, adj, None, {}, {}, empty_feats(adj))
        save(f"ggnn_layer_{agg}", x=x, y=y, **pack_adj(adj),
             **pack_specs([weights_from_reference_layer(layer)]))

    cases = {
        "mlp_layer_sum_target": dict(agg="sum"),
        "mlp_layer_max_target": dict(agg="max"),
        "mlp_layer_mean_notarget": dict(agg="mean", use_target_state_as_message_input=False),
        "mlp_layer_sum_hidden1": dict(agg="sum", mlp_hidden_layers=1),
        "mlp_layer_max_noln_nodense": dict(agg="max", use_layer_norm=False, use_dense_layer=False),
    }
    for i, (name, kw) in enumerate(cases.items()):
        torch.manual_seed(200 + i)
        kw = dict(kw)
        agg = kw.pop("agg")
        adj = tricky_adj(gen, n)
        x = torch.randn(n, H, generator=gen)
        layer = MlpMessagePassingLayer(H, 20, M, 3, agg, **kw).eval()
        # make LayerNorm affine / biases non-trivial so the fixture pins them
        with torch.no_grad():
            for p_name, p in layer.named_parameters():
                if "state_update" in p_name and p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
            y = layer(x, adj, None, {}, {}, empty_feats(adj))
        save(name, x=x, y=y, **pack_adj(adj), **pack_specs([weights_from_reference_layer(layer)]))


def training_gradients():
    """Backward pins: the reference's OWN layers in train mode (dropout 0 => deterministic) under torch
    autograd on CPU: output, d loss / d x and every parameter gradient for a fixed upstream gradient.
    Widths are multiples of 32 so the HIP edge form (grouped GEMM + weight-gradient kernel) can replay them."""
    gen = torch.Generator().manual_seed(777)
    n, H, M, T = 300, 32, 64, 4
    for name, make in (
            ("train_ggnn_max", lambda: GatedMessagePassingLayer(H, M, T, "max")),
            ("train_ggnn_sum", lambda: GatedMessagePassingLayer(H, M, T, "sum")),
            ("train_mlp_sum_target", lambda: MlpMessagePassingLayer(H, H, M, T, "sum")),
            ("train_mlp_max_notarget", lambda: MlpMessagePassingLayer(H, H, M, T, "max",
                                                                      use_target_state_as_message_input=False))):
        torch.manual_seed(len(name))
        adj = rand_adj(gen, n, [500, 0, 37, 260])
        x = torch.randn(n, H, generator=gen).requires_grad_(True)
        layer = make().train()
        with torch.no_grad():
            for p_name, p_ in layer.named_parameters():
                if "state_update" in p_name and p_.dim() == 1:
                    p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
        y = layer(x, adj, None, {}, {}, empty_feats(adj))
        gout = torch.randn(y.shape, generator=gen)
        y.backward(gout)
        grads = {"g.x": x.grad}
        for p_name, p_ in layer.named_parameters():
            grads["g." + p_name] = p_.grad
        save(name, x=x.detach(), y=y.detach(), gout=gout, **pack_adj(adj),
             **pack_specs([weights_from_reference_layer(layer)]), **grads)


class _Identity(torch.nn.Module):
    def forward(self, x):
        return x


def containers():
    gen = torch.Generator().manual_seed(4321)
    n, H = 60, 16
    node_to_graph = torch.repeat_interleave(torch.arange(3), 20)
    refs = {"supernodes": torch.tensor([0, 5, 21, 40, 59])}
    ref_g = {"supernodes": torch.tensor([0, 0, 1, 2, 2])}

    # Typilus GGNN architecture (typilus/train.py:39-65) at hidden 16, T0=3 -> T=7
    torch.manual_seed(7)
    T = 7
    ggnn = GatedMessagePassingLayer(H, H, T, "max", dropout_rate=0.1)
    r1 = ConcatResidualLayer(H)
    last = GatedMessagePassingLayer(2 * H, H, T, "max", dropout_rate=0.1)
    layers = [r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last]
    net = GraphNeuralNetwork(layers, _Identity(), introduce_backwards_edges=True,
                             add_self_edges=True).eval()
    adj = rand_adj(gen, n, [90, 0, 45])
    x = torch.randn(n, H, generator=gen)
    with torch.no_grad():
        out = net(node_data={"x": x}, adjacency_lists=[a for a in adj], edge_feature_data=[],
                  node_to_graph_idx=node_to_graph, reference_node_ids=refs,
                  reference_node_graph_idx=ref_g, num_graphs=3)
    s_ggnn, s_last = weights_from_reference_layer(ggnn), weights_from_reference_layer(last)
    specs = ([{"kind": "residual_origin", "name": "r1"}] + [s_ggnn] * 7
             + [{"kind": "residual_concat", "name": "r1"}, s_last])
    assert out.node_to_graph_idx is node_to_graph and out.node_idx_references is refs
    save("gnn_stack_ggnn_typilus", x=x, y=out.output_node_representations,
         num_edges=np.asarray(net.report_metrics()["num_edges"]),
         node_to_graph_idx=node_to_graph, **pack_adj(adj), **pack_specs(specs))

    # VarMisuse-style MLP architecture (varmisuse/train.py:42-74) at hidden 16, T0=2 -> T=5
    torch.manual_seed(8)
    T = 5
    mk = lambda: MlpMessagePassingLayer(H, H, H, T, "max", dropout_rate=0.1)  # noqa: E731
    mk2 = lambda: MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max", dropout_rate=0.1)  # noqa: E731
    r1, r2 = ConcatResidualLayer(H), MeanResidualLayer(H)
    mods = [r1.pass_through_dummy_layer(), mk(), mk(), r1, mk2(),
            r2.pass_through_dummy_layer(), mk(), r2]
    net = GraphNeuralNetwork(mods, _Identity(), introduce_backwards_edges=True,
                             add_self_edges=True).eval()
    adj = rand_adj(gen, n, [70, 31])
    x = torch.randn(n, H, generator=gen)
    with torch.no_grad():
        out = net(node_data={"x": x}, adjacency_lists=[a for a in adj], edge_feature_data=[],
                  node_to_graph_idx=node_to_graph, reference_node_ids=refs,
                  reference_node_graph_idx=ref_g, num_graphs=3)
    specs = []
    for m in mods:
        k = type(m).__name__
        if k == "_ResidualOriginLayer":
            specs.append({"kind": "residual_origin", "name": "r1" if not specs else "r2"})
        elif k == "ConcatResidualLayer":
            specs.append({"kind": "residual_concat", "name": "r1"})
        elif k == "MeanResidualLayer":
            specs.append({"kind": "residual_mean", "name": "r2"})
        else:
            specs.append(weights_from_reference_layer(m))
    save("gnn_stack_mlp_varmisuse", x=x, y=out.output_node_representations,
         num_edges=np.asarray(net.report_metrics()["num_edges"]),
         node_to_graph_idx=node_to_graph, **pack_adj(adj), **pack_specs(specs))


def varmisuse_ggnn():
    """VarMisuse GGNN architecture (varmisuse/train.py:76-107) at hidden 16, T0=2 -> T=5, with one
    weighted-sum and one max global exchange."""
    gen = torch.Generator().manual_seed(99)
    n, H, T = 60, 16, 5
    node_to_graph = torch.repeat_interleave(torch.arange(3), torch.tensor([25, 5, 30]))
    torch.manual_seed(12)
    ggnn = GatedMessagePassingLayer(H, H, T, "sum", dropout_rate=0.01)
    r1, r2 = MeanResidualLayer(H), MeanResidualLayer(H)
    g1 = GruGlobalStateUpdate(WeightedSumVarSizedElementReduce(H), H, H, dropout_rate=0.1)
    g2 = GruGlobalStateUpdate(SimpleVarSizedElementReduce("max"), H, H, dropout_rate=0.1)
    mods = [r1.pass_through_dummy_layer(), r2.pass_through_dummy_layer(), ggnn, ggnn, ggnn, g1, ggnn, r1,
            ggnn, ggnn, ggnn, g2, ggnn, r2]
    net = GraphNeuralNetwork(mods, _Identity(), introduce_backwards_edges=True, add_self_edges=True).eval()
    adj = rand_adj(gen, n, [80, 33])
    x = torch.randn(n, H, generator=gen)
    with torch.no_grad():
        out = net(node_data={"x": x}, adjacency_lists=[a for a in adj], edge_feature_data=[],
                  node_to_graph_idx=node_to_graph, reference_node_ids={}, reference_node_graph_idx={},
                  num_graphs=3)
    s_ggnn = weights_from_reference_layer(ggnn)
    sg1, sg2 = weights_from_reference_layer(g1), weights_from_reference_layer(g2)
    specs = [{"kind": "residual_origin", "name": "r1"}, {"kind": "residual_origin", "name": "r2"},
             s_ggnn, s_ggnn, s_ggnn, sg1, s_ggnn, {"kind": "residual_mean", "name": "r1"},
             s_ggnn, s_ggnn, s_ggnn, sg2, s_ggnn, {"kind": "residual_mean", "name": "r2"}]
    save("gnn_stack_ggnn_varmisuse_global", x=x, y=out.output_node_representations,
         node_to_graph_idx=node_to_graph, **pack_adj(adj), **pack_specs(specs))


class _NodeModel(AbstractNeuralModel):
    """Minimal node 'embedder' model: a node is an int id, the minibatch is the id list."""

    def initialize_metadata(self):
        pass

    def update_metadata_from(self, datapoint):
        pass

    def finalize_metadata(self):
        pass

    def build_neural_module(self):
        return _Identity()

    def tensorize(self, datapoint):
        return int(datapoint)

    def initialize_minibatch(self):
        return {"ids": []}

    def extend_minibatch_with(self, tensorized_datapoint, partial_minibatch):
        partial_minibatch["ids"].append(tensorized_datapoint)
        return True

    def finalize_minibatch(self, accumulated_minibatch_data, device):
        return {"x": torch.tensor(accumulated_minibatch_data["ids"], dtype=torch.int64)}


def batcher():
    rng = np.random.RandomState(1234)
    edge_names = ["child", "next", "uses"]
    graphs = []
    for g in range(7):
        nn_ = int(rng.randint(3, 12))
        edges = {}
        for name in edge_names:
            if name == "uses" and g % 3 == 0:
                continue  # a graph without one edge type (:318-323 zero-length branch)
            ne = int(rng.randint(1, 3 * nn_))
            edges[name] = [(int(a), int(b)) for a, b in rng.randint(0, nn_, size=(ne, 2))]
        refs = {"supernodes": [int(v) for v in rng.choice(nn_, size=2, replace=False)],
                "slot": [int(rng.randint(0, nn_))]}
        graphs.append(GraphData(node_information=list(range(nn_)), edges=edges, reference_nodes=refs))

    model = GraphNeuralNetworkModel(
        node_representation_model=_NodeModel(),
        message_passing_layer_creator=lambda n: [GatedMessagePassingLayer(4, 4, n, "sum")],
        stop_extending_minibatch_after_num_nodes=20, add_self_edges=True)
    model.compute_metadata(iter(graphs), parallelize=False)
    order = sorted(edge_names, key=model.edge_idx_by_name)
    arrays = {"edge_type_order": np.asarray(order), "num_graphs_in": np.asarray(len(graphs)),
              "stop_after": np.asarray(20)}
    tensorized = [model.tensorize(g) for g in graphs]
    for gi, t in enumerate(tensorized):
        arrays[f"g{gi}.num_nodes"] = np.asarray(t.num_nodes)
        for ti, (s, d) in enumerate(t.adjacency_lists):
            arrays[f"g{gi}.adj.{ti}.src"], arrays[f"g{gi}.adj.{ti}.dst"] = s, d
        for k, v in t.reference_nodes.items():
            arrays[f"g{gi}.ref.{k}"] = v
    # replay the reference's own minibatch loop (abstractneuralmodel.py:290-319)
    mbs, mb = [], model.initialize_minibatch()
    for t in tensorized:
        more = model.extend_minibatch_with(t, mb)
        if not more:
            mbs.append(model.finalize_minibatch(mb, "cpu"))
            mb = model.initialize_minibatch()
    if mb["num_nodes_per_graph"]:
        mbs.append(model.finalize_minibatch(mb, "cpu"))
    arrays["num_minibatches"] = np.asarray(len(mbs))
    for bi, fin in enumerate(mbs):
        arrays[f"mb{bi}.num_graphs"] = np.asarray(fin["num_graphs"])
        arrays[f"mb{bi}.node_to_graph_idx"] = fin["node_to_graph_idx"].numpy()
        for ti, (s, d) in enumerate(fin["adjacency_lists"]):
            arrays[f"mb{bi}.adj.{ti}.src"], arrays[f"mb{bi}.adj.{ti}.dst"] = s.numpy(), d.numpy()
        for k in fin["reference_node_ids"]:
            arrays[f"mb{bi}.ref_ids.{k}"] = fin["reference_node_ids"][k].numpy()
            arrays[f"mb{bi}.ref_gidx.{k}"] = fin["reference_node_graph_idx"][k].numpy()
    save("batcher", **arrays)


def wide_layers():
    """Round 4: fixtures at the widths the SHIPPED fast kernels take (K % 64 == 0: k_stream_gru, k_stream_edge incl. the
    shared-row form, k_stream_linear, k_wgrad_stream), so that those kernels are compared with the reference's own
    output directly and not only through the oracle.  Own generators: the fixtures above stay bit-identical."""
    gen = torch.Generator().manual_seed(20260926)
    n, H = 600, 128
    # one GGNN and one MLP-MP layer at H = M = 128 on the tricky adjacency (gatedmessagepassing.py:37-69,
    # mlpmessagepassing.py:68-117)
    torch.manual_seed(501)
    adj = tricky_adj(gen, n)
    x = torch.randn(n, H, generator=gen)
    layer = GatedMessagePassingLayer(H, H, 3, "max").eval()
    with torch.no_grad():
        y = layer(x, adj, None, {}, {}, empty_feats(adj))
    save("ggnn_layer_max_w128", x=x, y=y, **pack_adj(adj), **pack_specs([weights_from_reference_layer(layer)]))

    torch.manual_seed(502)
    adj = tricky_adj(gen, n)
    x = torch.randn(n, H, generator=gen)
    layer = MlpMessagePassingLayer(H, H, H, 3, "sum").eval()
    with torch.no_grad():
        for p_name, p in layer.named_parameters():
            if "state_update" in p_name and p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
        y = layer(x, adj, None, {}, {}, empty_feats(adj))
    save("mlp_layer_sum_target_w128", x=x, y=y, **pack_adj(adj), **pack_specs([weights_from_reference_layer(layer)]))

    # Typilus GGNN architecture (typilus/train.py:39-65) at hidden 64, T0 = 3 -> T = 7, three graphs of 400 nodes; few
    # enough edges per type that the per-edge form with shared message rows is what the layers choose (E 1.25 < N T)
    n, H, T = 1200, 64, 7
    node_to_graph = torch.repeat_interleave(torch.arange(3), 400)
    refs = {"supernodes": torch.tensor([0, 5, 421, 840, 1199])}
    ref_g = {"supernodes": torch.tensor([0, 0, 1, 2, 2])}
    torch.manual_seed(503)
    ggnn = GatedMessagePassingLayer(H, H, T, "max", dropout_rate=0.1)
    r1 = ConcatResidualLayer(H)
    last = GatedMessagePassingLayer(2 * H, H, T, "max", dropout_rate=0.1)
    layers = [r1.pass_through_dummy_layer()] + [ggnn] * 7 + [r1, last]
    net = GraphNeuralNetwork(layers, _Identity(), introduce_backwards_edges=True, add_self_edges=True).eval()
    adj = []
    for c in (1300, 0, 900):        # edges stay inside their graph, like a real disjoint-union batch
        gidx = torch.randint(0, 3, (c,), generator=gen)
        s = torch.randint(0, 400, (c,), generator=gen, dtype=torch.int64) + 400 * gidx
        d = torch.randint(0, 400, (c,), generator=gen, dtype=torch.int64) + 400 * gidx
        adj.append((s, d))
    x = torch.randn(n, H, generator=gen)
    with torch.no_grad():
        out = net(node_data={"x": x}, adjacency_lists=[a for a in adj], edge_feature_data=[],
                  node_to_graph_idx=node_to_graph, reference_node_ids=refs,
                  reference_node_graph_idx=ref_g, num_graphs=3)
    s_ggnn, s_last = weights_from_reference_layer(ggnn), weights_from_reference_layer(last)
    specs = ([{"kind": "residual_origin", "name": "r1"}] + [s_ggnn] * 7
             + [{"kind": "residual_concat", "name": "r1"}, s_last])
    save("gnn_stack_ggnn_typilus_w64", x=x, y=out.output_node_representations,
         num_edges=np.asarray(net.report_metrics()["num_edges"]),
         node_to_graph_idx=node_to_graph, **pack_adj(adj), **pack_specs(specs))

    # training gradients of the reference's own layers at H = M = 64 (and one GGNN at 128, the width of the streaming
    # weight-gradient kernel's dropout / 128-column tiles)
    n, T = 300, 4
    for name, H, make in (
            ("train_ggnn_max_w64", 64, lambda: GatedMessagePassingLayer(64, 64, T, "max")),
            ("train_mlp_sum_target_w64", 64, lambda: MlpMessagePassingLayer(64, 64, 64, T, "sum")),
            ("train_ggnn_sum_w128", 128, lambda: GatedMessagePassingLayer(128, 128, T, "sum"))):
        torch.manual_seed(600 + len(name))
        adj = rand_adj(gen, n, [500, 0, 37, 260])
        x = torch.randn(n, H, generator=gen).requires_grad_(True)
        layer = make().train()
        with torch.no_grad():
            for p_name, p_ in layer.named_parameters():
                if "state_update" in p_name and p_.dim() == 1:
                    p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
        y = layer(x, adj, None, {}, {}, empty_feats(adj))
        gout = torch.randn(y.shape, generator=gen)
        y.backward(gout)
        grads = {"g.x": x.grad}
        for p_name, p_ in layer.named_parameters():
            grads["g." + p_name] = p_.grad
        save(name, x=x.detach(), y=y.detach(), gout=gout, **pack_adj(adj),
             **pack_specs([weights_from_reference_layer(layer)]), **grads)


if __name__ == "__main__":
    torch.set_num_threads(1)
    only = sys.argv[1:]
    for fn in (single_layers, training_gradients, containers, varmisuse_ggnn, batcher, wide_layers):
        if not only or fn.__name__ in only:
            fn()
