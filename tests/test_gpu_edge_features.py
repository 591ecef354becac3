"""Edge features on the hot path (gatedmessagepassing.py:54-61, mlpmessagepassing.py:90-98, the features
arriving through graphneuralnetwork.py:162-186): at inference the grouped per-edge GEMM gathers
[x[src] | x[dst] | features[e]] itself (`ptgnn_amd_edge_linear_feat_f32`) -- checked against the CPU oracle, with the
kernel that ran asserted through the launch counters and torch's row gathers forbidden."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import to_cuda_adj  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _graph(n, counts, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randint(0, n, (c,), generator=g), torch.randint(0, n, (c,), generator=g)) for c in counts], g


@pytest.mark.parametrize("H,M,F,use_dst,act", [(64, 64, 8, False, None), (128, 128, 16, False, None),
                                               (32, 24, 5, False, "tanh"), (64, 128, 3, True, "relu"),
                                               (96, 36, 40, True, None), (32, 64, 1, False, None)])
def test_edge_linear_with_feature_rows_matches_float64(H, M, F, use_dst, act):
    """Op level: ragged types (one empty, one of a single edge, one past a tile boundary), feature widths below 4, not a
    multiple of 4 and wider than a K chunk; against the concatenated-input GEMM in float64."""
    from ptgnn_amd import ops
    n = 700
    adj, g = _graph(n, (1300, 0, 1, 129, 517), seed=H + F)
    x = torch.randn(n, H, generator=g)
    Hs = H * (2 if use_dst else 1)
    ws = [torch.randn(M, Hs + F, generator=g) / (Hs + F) ** 0.5 for _ in adj]
    feats = [torch.randn(int(a[0].shape[0]), F, generator=g) for a in adj]
    want = []
    for (s, d), f, w in zip(adj, feats, ws):
        inp = [x[s].double()] + ([x[d].double()] if use_dst else []) + [f.double()]
        y = torch.cat(inp, -1) @ w.double().t()
        want.append(torch.tanh(y) if act == "tanh" else (torch.relu(y) if act == "relu" else y))
    want = torch.cat(want)
    before = ops.launch_counts()
    got = ops.edge_linear(x.cuda(), to_cuda_adj(adj), [w.cuda() for w in ws], use_dst, act=act,
                          edge_feats=[f.cuda() for f in feats])
    ran = ops.launches_since(before)
    assert ran == {"k_edge_linear": 1}, ran
    assert got.shape == want.shape
    assert float((got.cpu().double() - want).abs().max()) <= TOL


def test_edge_linear_feature_argument_checks():
    from ptgnn_amd import _lib, ops
    adj, g = _graph(50, (40,), seed=1)
    x = torch.randn(50, 32, generator=g).cuda()
    w = torch.randn(16, 40, generator=g).cuda()
    f = torch.randn(40, 8, generator=g).cuda()
    with pytest.raises(_lib.PtgnnAmdError, match="does not match"):
        ops.edge_linear(x, to_cuda_adj(adj), [w[:, :36].contiguous()], False, edge_feats=[f])
    with pytest.raises(_lib.PtgnnAmdError, match="features of type 0"):
        ops.edge_linear(x, to_cuda_adj(adj), [w], False, edge_feats=[f[:39]])
    with pytest.raises(_lib.PtgnnAmdError, match="dropout"):
        ops.edge_linear(x, to_cuda_adj(adj), [w], False, dropout=(1, 0.1, 7), edge_feats=[f])


@pytest.mark.parametrize("kind", ["ggnn_sum", "ggnn_max", "mlp_src", "mlp_dst_hidden"])
def test_layers_with_edge_features_take_the_fused_gather(kind, monkeypatch):
    """Layer level vs the oracle (mp_oracle.ggnn_layer / mlp_mp_layer with features): no torch row gather, no [E, H + F]
    concat -- `index_select` and `cat` of 2-D per-edge inputs are forbidden while the layer runs."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops
    n, H, M, F, T = 900, 64, 64, 12, 4
    adj, g = _graph(n, (2100, 700, 0, 333), seed=23)
    x = torch.randn(n, H, generator=g)
    feats = [torch.randn(int(a[0].shape[0]), F, generator=g) for a in adj]
    torch.manual_seed(24)
    if kind.startswith("ggnn"):
        layer, fn = L.GatedMessagePassingLayer(H, M, T, kind.split("_")[1], edge_feature_dimension=F), O.ggnn_layer
    elif kind == "mlp_src":
        layer = L.MlpMessagePassingLayer(H, H, M, T, "max", mlp_hidden_layers=0, features_dimension=F,
                                         use_target_state_as_message_input=False)
        fn = O.mlp_mp_layer
    else:
        layer = L.MlpMessagePassingLayer(H, 48, M, T, "sum", mlp_hidden_layers=1, features_dimension=F)
        fn = O.mlp_mp_layer
    want = fn(x, adj, feats, layer.export_weights())
    layer = layer.cuda().eval()
    cadj, cfeats, xc = to_cuda_adj(adj), [f.cuda() for f in feats], x.cuda()
    ops.clear_plan_cache()
    real_select = torch.Tensor.index_select

    def no_select(self, *a, **k):
        raise AssertionError("torch row gather on the feature path")
    monkeypatch.setattr(torch.Tensor, "index_select", no_select)
    before = ops.launch_counts()
    with torch.no_grad():
        got = layer(xc, cadj, None, {}, {}, cfeats)
    ran = ops.launches_since(before)
    monkeypatch.setattr(torch.Tensor, "index_select", real_select)
    assert ran.get("k_edge_linear", 0) == 1, ran
    assert float((got.cpu() - want).abs().max()) <= TOL


def test_container_feeds_embedded_edge_features_to_the_fused_gather():
    """graphneuralnetwork.py:162-186 end to end at widths the fused form takes (H = 32, F = 8): reverse edges reuse the
    forward features, self edges get zeros; two GGNN layers; vs oracle.gnn_forward."""
    from oracle import mp_oracle as O
    from ptgnn_amd import layers as L, ops, workloads
    from ptgnn_amd.gnn import GraphNeuralNetwork
    H, F = 32, 8
    mb = workloads.batched_graphs(4, 250, 3, 2.0, seed=31)
    N = mb["num_nodes"]

    class Embed(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(6, F, bias=False)

        def forward(self, features):
            return self.lin(features)

    torch.manual_seed(32)
    emb = Embed()
    T = 2 * len(mb["adjacency_lists"]) + 1
    layers = [L.GatedMessagePassingLayer(H, H, T, "sum", edge_feature_dimension=F) for _ in range(2)]
    gen = torch.Generator().manual_seed(33)
    raw = [torch.randn(int(s.shape[0]), 6, generator=gen) for s, _ in mb["adjacency_lists"]]
    x = workloads.node_states(N, H, seed=6)
    feats = [emb.lin(r).detach() for r in raw]
    want, _ = O.gnn_forward(x, mb["adjacency_lists"], [l.export_weights() for l in layers], True, True,
                            edge_features=feats)
    net = GraphNeuralNetwork(layers, torch.nn.Identity(), True, True, edge_feature_embedder=emb).cuda().eval()
    before = ops.launch_counts()
    with torch.no_grad():
        out = net(node_data={"input": x.cuda()}, adjacency_lists=to_cuda_adj(mb["adjacency_lists"]),
                  edge_feature_data=[{"features": r.cuda()} for r in raw],
                  node_to_graph_idx=mb["node_to_graph_idx"].cuda(), reference_node_ids={},
                  reference_node_graph_idx={}, num_graphs=mb["num_graphs"])
    ran = ops.launches_since(before)
    assert ran.get("k_edge_linear", 0) == 2, ran
    assert float((out.output_node_representations.cpu() - want).abs().max()) <= TOL


@pytest.mark.parametrize("kind,use_target,hidden", [("ggnn", False, 0), ("mlp", True, 0), ("mlp", False, [32])])
@pytest.mark.parametrize("F", [8, 5])
def test_training_with_edge_features_runs_on_the_grouped_gemm_and_matches_the_cpu_route(kind, use_target, hidden, F, monkeypatch):
    """Round 5: training with per-edge features is ONE autograd node around the grouped per-edge GEMM
    (scatter._EdgeLinearFeat: no index_select, no [E, H + F] concat).  Output, d x, d features and every parameter gradient
    against the same layer on CPU tensors (ptgnn_amd/torch_route.py = the reference's arithmetic under torch autograd)."""
    import copy
    import numpy as np
    from ptgnn_amd import layers as L, ops
    n, H, M = 900, 64, 64
    adj, g = _graph(n, (2100, 0, 1, 777), seed=3 + F)
    T = len(adj)
    torch.manual_seed(5)
    if kind == "ggnn":
        cpu_layer = L.GatedMessagePassingLayer(H, M, T, "max", edge_feature_dimension=F)
    else:
        cpu_layer = L.MlpMessagePassingLayer(H, H, M, T, "sum", use_target_state_as_message_input=use_target,
                                             mlp_hidden_layers=hidden, features_dimension=F)
    gpu_layer = copy.deepcopy(cpu_layer).cuda()
    x = torch.randn(n, H, generator=g)
    feats = [torch.randn(int(a[0].shape[0]), F, generator=g) for a in adj]
    xc = x.clone().requires_grad_(True)
    fc = [f.clone().requires_grad_(True) for f in feats]
    yc = cpu_layer.train()(xc, adj, None, {}, {}, fc)
    gout = torch.linspace(-1, 1, yc.numel()).view_as(yc)
    yc.backward(gout)
    xg = x.cuda().requires_grad_(True)
    fg = [f.cuda().requires_grad_(True) for f in feats]
    cadj = to_cuda_adj(adj)
    ops.clear_plan_cache()

    def no_gather(*a, **k):
        raise AssertionError("index_select on the per-edge path")
    before = ops.launch_counts()
    monkeypatch.setattr(torch.Tensor, "index_select", no_gather)
    yg = gpu_layer.train()(xg, cadj, None, {}, {}, fg)
    monkeypatch.undo()
    assert ops.launches_since(before).get("k_edge_linear", 0) >= 1
    yg.backward(gout.cuda())
    tol = lambda t: 5e-5 * max(1.0, float(t.abs().max()))  # noqa: E731
    np.testing.assert_allclose(yg.detach().cpu().numpy(), yc.detach().numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=0, atol=tol(xc.grad))
    for a, b in zip(fg, fc):
        if b.shape[0]:
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=tol(b.grad))
    for (k, pc), (_, pg) in zip(cpu_layer.named_parameters(), gpu_layer.named_parameters()):
        if pc.grad is None:
            assert pg.grad is None or float(pg.grad.abs().max()) == 0.0, k
            continue
        np.testing.assert_allclose(pg.grad.cpu().numpy(), pc.grad.numpy(), rtol=0, atol=tol(pc.grad), err_msg=k)
