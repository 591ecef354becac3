"""Pins the CPU restatement (oracle/mp_oracle.py) against golden vectors produced by the
reference's OWN modules (tests/golden/make_golden.py, run in the authoring container)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import mp_oracle as O
from oracle.fixtures import unpack_adj, unpack_specs

LAYER_CASES = ["ggnn_layer_sum", "ggnn_layer_mean", "ggnn_layer_max", "ggnn_layer_min",
               "mlp_layer_sum_target", "mlp_layer_max_target", "mlp_layer_mean_notarget",
               "mlp_layer_sum_hidden1", "mlp_layer_max_noln_nodense",
               # round 4: the widths the shipped streaming kernels take (K % 64 == 0)
               "ggnn_layer_max_w128", "mlp_layer_sum_target_w128"]
TOL = 2e-6  # explicit-formula LayerNorm/GRU vs torch's fused CPU kernels


@pytest.mark.parametrize("name", LAYER_CASES)
def test_single_layer_matches_reference(name):
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)
    x = torch.from_numpy(g["x"])
    y = O.run_layer_stack(x, adj, [spec])
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=TOL)
    # fp64 evaluation of the same spec: error attribution, budget is 1e-5
    y64 = O.run_layer_stack(x.double(), adj, [O.cast_spec(spec, torch.float64)])
    assert np.abs(y64.numpy() - g["y"]).max() < 1e-5


@pytest.mark.parametrize("name,bwd,selfe", [("gnn_stack_ggnn_typilus", True, True),
                                            ("gnn_stack_mlp_varmisuse", True, True),
                                            ("gnn_stack_ggnn_typilus_w64", True, True)])
def test_container_matches_reference(name, bwd, selfe):
    g = load_golden(name)
    adj, specs = unpack_adj(g), unpack_specs(g)
    x = torch.from_numpy(g["x"])
    n_types_before = len(adj)
    y, num_edges = O.gnn_forward(x, adj, specs, bwd, selfe)
    assert len(adj) == n_types_before, "oracle must not mutate the caller's list"
    assert num_edges == int(g["num_edges"])           # graphneuralnetwork.py:198
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-5)


def test_varmisuse_ggnn_stack_with_global_exchange_matches_reference():
    g = load_golden("gnn_stack_ggnn_varmisuse_global")
    adj, specs = unpack_adj(g), unpack_specs(g)
    x = torch.from_numpy(g["x"])
    n2g = torch.from_numpy(g["node_to_graph_idx"])
    y, _ = O.gnn_forward(x, adj, specs, True, True, node_to_graph_idx=n2g)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-5)


def test_augmentation_order():
    a = [(torch.tensor([0, 1]), torch.tensor([1, 2])), (torch.tensor([2]), torch.tensor([0]))]
    aug = O.augment_adjacency(a, 3, True, True)
    assert len(aug) == 5
    assert aug[2][0].tolist() == [1, 2] and aug[2][1].tolist() == [0, 1]   # reversed type 0
    assert aug[3][0].tolist() == [0] and aug[3][1].tolist() == [2]         # reversed type 1
    assert aug[4][0].tolist() == [0, 1, 2] == aug[4][1].tolist()           # self edges last


def test_batcher_bit_exact():
    g = load_golden("batcher")
    T0 = len(g["edge_type_order"])
    graphs = []
    for gi in range(int(g["num_graphs_in"])):
        refs = {k.split(".")[-1]: g[k] for k in g.files if k.startswith(f"g{gi}.ref.")}
        graphs.append({"num_nodes": int(g[f"g{gi}.num_nodes"]),
                       "adjacency_lists": [(g[f"g{gi}.adj.{t}.src"], g[f"g{gi}.adj.{t}.dst"])
                                           for t in range(T0)],
                       "reference_nodes": refs})
    mbs = list(O.batch_graphs(graphs, T0, int(g["stop_after"])))
    assert len(mbs) == int(g["num_minibatches"])
    for bi, mb in enumerate(mbs):
        assert mb["num_graphs"] == int(g[f"mb{bi}.num_graphs"])
        assert mb["node_to_graph_idx"].dtype == torch.int64
        np.testing.assert_array_equal(mb["node_to_graph_idx"].numpy(), g[f"mb{bi}.node_to_graph_idx"])
        for t in range(T0):
            s, d = mb["adjacency_lists"][t]
            assert s.dtype == torch.int64 and d.dtype == torch.int64
            np.testing.assert_array_equal(s.numpy(), g[f"mb{bi}.adj.{t}.src"])
            np.testing.assert_array_equal(d.numpy(), g[f"mb{bi}.adj.{t}.dst"])
        for k in mb["reference_node_ids"]:
            np.testing.assert_array_equal(mb["reference_node_ids"][k].numpy(), g[f"mb{bi}.ref_ids.{k}"])
            np.testing.assert_array_equal(mb["reference_node_graph_idx"][k].numpy(),
                                          g[f"mb{bi}.ref_gidx.{k}"])


TRAIN_CASES = ["train_ggnn_max", "train_ggnn_sum", "train_mlp_sum_target", "train_mlp_max_notarget",
               "train_ggnn_max_w64", "train_mlp_sum_target_w64", "train_ggnn_sum_w128"]


def _grad_pairs(spec, g):
    """(oracle spec tensor, golden gradient) pairs: the spec layout <-> the reference's parameter names."""
    if spec["kind"] == "ggnn":
        p = "g._GatedMessagePassingLayer__"
        pairs = [(w, g[f"{p}edge_message_transformation_layers.{t}.weight"]) for t, w in enumerate(spec["edge_w"])]
        pairs += [(spec[k], g[f"{p}state_update.{n}"]) for k, n in
                  (("w_ih", "weight_ih"), ("w_hh", "weight_hh"), ("b_ih", "bias_ih"), ("b_hh", "bias_hh"))]
        return pairs
    p = "g._MlpMessagePassingLayer__"
    pairs = [(ws[0], g[f"{p}edge_message_transformation_layers.{t}._MLP__mlp_modules.1.weight"])
             for t, ws in enumerate(spec["edge_mlp"])]
    pairs += [(spec[k], g[f"{p}state_update.{n}"]) for k, n in
              (("ln_w", "0.weight"), ("ln_b", "0.bias"), ("dense_w", "1.weight"), ("dense_b", "1.bias"))]
    return pairs


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_oracle_autograd_matches_reference_gradients(name):
    """Backward pin: torch autograd through the oracle restatement reproduces the gradients the reference's
    own layers produced (tests/golden/make_golden.py::training_gradients)."""
    g = load_golden(name)
    adj, (spec,) = unpack_adj(g), unpack_specs(g)

    def req(v):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            return v.clone().requires_grad_(True)
        if isinstance(v, list):
            return [req(u) for u in v]
        return v
    spec = {k: req(v) for k, v in spec.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = O.run_layer_stack(x, adj, [spec])
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=0, atol=TOL)
    y.backward(torch.from_numpy(g["gout"]))
    np.testing.assert_allclose(x.grad.numpy(), g["g.x"], rtol=0, atol=1e-5 * max(1.0, np.abs(g["g.x"]).max()))
    for t, want in _grad_pairs(spec, g):
        np.testing.assert_allclose(t.grad.numpy(), want, rtol=0, atol=1e-5 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("kind,agg", [("ggnn", "sum"), ("ggnn", "max"), ("mlp", "sum"), ("mlp", "mean")])
def test_row_restricted_oracle_equals_the_whole_graph_oracle(kind, agg):
    """oracle/mp_oracle.py `layer_on_rows` + `row_chunks` (the every-row checker of the config-5 shard,
    oracle/fullrow.py) against the whole-graph restatement that the reference fixtures pin: chunks of destination rows,
    concatenated, and an arbitrary row subset -- same arithmetic up to BLAS blocking (<= 1e-6)."""
    import torch
    from oracle import fullrow, mp_oracle as O
    from ptgnn_amd import layers as L
    torch.manual_seed(0)
    N, H, T = 3000, 32, 3
    adj = [(torch.randint(0, N, (e,)), torch.randint(0, N // 3, (e,))) for e in (5000, 100, 9000)]
    x = torch.randn(N, H)
    deg = torch.bincount(torch.cat([d for _, d in adj]), minlength=N)
    layer = (L.GatedMessagePassingLayer(H, 48, T, agg) if kind == "ggnn"
             else L.MlpMessagePassingLayer(H, H, 40, T, agg, use_target_state_as_message_input=agg == "sum"))
    spec = layer.export_weights()
    feats = [torch.empty(a[0].shape[0], 0) for a in adj]
    with torch.no_grad():
        want = (O.ggnn_layer if kind == "ggnn" else O.mlp_mp_layer)(x, adj, feats, spec)
        chunks = list(O.row_chunks(deg, 1500))
        got = torch.cat([O.layer_on_rows(x, adj, spec, torch.arange(lo, hi)) for lo, hi in chunks])
        rows = torch.tensor([0, 5, 17, 999, 2999])
        some = O.layer_on_rows(x, adj, spec, rows)
    assert len(chunks) > 5 and chunks[0][0] == 0 and chunks[-1][1] == N
    assert all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    assert float((got - want).abs().max()) <= 1e-6 and float((some - want[rows]).abs().max()) <= 1e-6
    res = fullrow.full_row_parity(spec, adj, x, want.clone(), max_edges=2000)
    assert res["ok"] and res["strict_1e-5"] and res["rows_checked"] == N and res["edges_checked"] == 14100
    bad = want.clone()
    bad[17] += 1e-4
    assert not fullrow.full_row_parity(spec, adj, x, bad, max_edges=2000)["ok"]
