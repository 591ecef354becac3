"""TEST INFRASTRUCTURE ONLY -- never imported by the product package `ptgnn_amd`.

CPU restatement (torch-CPU fp32/fp64 + numpy for the integer batching) of the one hot path of
microsoft/ptgnn that `ptgnn_amd` accelerates.  Every function cites the reference lines it
follows (paths relative to /root/reference).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module, and only as the checker / the reported
CPU baseline -- never as the thing measured or shipped.

Layer weights are plain dicts of torch tensors so that the same oracle can be fed from the
reference's own modules (tests/golden/make_golden.py), from golden ``.npz`` fixtures, or from
`ptgnn_amd` layers (``layer.export_weights()``).

PARITY STATUS: the reference's test-suite holds no fixtures for this path, so
this restatement is pinned against outputs of the reference's *own* modules executed in the
authoring container (fixtures in tests/golden/*.npz, generator tests/golden/make_golden.py),
on top of the restated third-party `torch_scatter` (oracle/scatter_ref.py).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import math
import numpy as np
import torch

from oracle.scatter_ref import scatter

Adj = List[Tuple[torch.Tensor, torch.Tensor]]


# --------------------------------------------------------------------------------------------
# dense building blocks (torch.nn definitions, spelled out)
# --------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """nn.Linear: y = x W^T + b  (weight is [out, in])."""
    y = x @ weight.t()
    return y if bias is None else y + bias


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell (gate order r, z, n), as used at gatedmessagepassing.py:25,69.

    r = sigmoid(W_ir x + b_ir + W_hr h + b_hr); z likewise;
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h
    """
    H = h.shape[1]
    gi = linear(x, w_ih, b_ih)
    gh = linear(h, w_hh, b_hh)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def gelu(x):
    """nn.GELU() exact-erf form (mlpmessagepassing.py:20)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, weight, bias, eps: float = 1e-5):
    """nn.LayerNorm over the last dim, biased variance (mlpmessagepassing.py:58)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def mlp(x, weights: Sequence[torch.Tensor]):
    """ptgnn/neuralmodels/mlp.py:50-80 in eval mode: bias-free Linear stack, ReLU between
    hidden layers, no activation after the last layer."""
    for w in weights[:-1]:
        x = torch.relu(linear(x, w))
    return linear(x, weights[-1])


# --------------------------------------------------------------------------------------------
# the scatter seam
# --------------------------------------------------------------------------------------------
def aggregate_messages(messages, message_targets, num_nodes: int, aggregation_fn: str):
    """abstractmessagepassing.py:38-50: fp32 up-cast -> torch_scatter.scatter -> cast back."""
    dt = messages.dtype
    up = messages if dt == torch.float64 else messages.to(torch.float32)
    return scatter(up, index=message_targets, dim=0, dim_size=num_nodes,
                   reduce=aggregation_fn).to(dt)


# --------------------------------------------------------------------------------------------
# message passing layers (eval mode: every nn.Dropout is the identity)
# --------------------------------------------------------------------------------------------
def ggnn_layer(node_states, adjacency_lists: Adj, edge_features, w: Dict, return_aggregate=False):
    """GatedMessagePassingLayer.forward, gatedmessagepassing.py:37-69.

    w: {"edge_w": [T x [M, H+F]], "w_ih": [3H, M], "w_hh": [3H, H], "b_ih", "b_hh", "agg": str}
    """
    assert len(adjacency_lists) == len(w["edge_w"])                               # :47
    message_targets = torch.cat([adj[1] for adj in adjacency_lists])              # :46
    all_messages = []
    for (src, _dst), feats, w_t in zip(adjacency_lists, edge_features, w["edge_w"]):  # :50-61
        edge_source_states = node_states.index_select(0, src)                     # :54-56
        all_messages.append(linear(torch.cat([edge_source_states, feats], -1), w_t))  # :57-61
    aggregated = aggregate_messages(torch.cat(all_messages, 0), message_targets,
                                    node_states.shape[0], w["agg"])               # :63-68
    out = gru_cell(aggregated, node_states, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"])  # :69
    return (out, aggregated) if return_aggregate else out


def mlp_mp_layer(node_states, adjacency_lists: Adj, edge_features, w: Dict,
                 return_aggregate=False):
    """MlpMessagePassingLayer.forward, mlpmessagepassing.py:68-117.

    w: {"edge_mlp": [T x [list of Linear weights]], "use_target": bool, "agg": str,
        "gelu": bool, "ln_w"/"ln_b" (optional), "dense_w"/"dense_b" (optional), "tanh": bool}
    """
    assert len(adjacency_lists) == len(w["edge_mlp"])                             # :77-79
    all_targets, all_messages = [], []
    for (src, dst), feats, mlp_w in zip(adjacency_lists, edge_features, w["edge_mlp"]):
        all_targets.append(dst)                                                   # :86
        message_input = node_states.index_select(0, src)                          # :88
        if w["use_target"]:
            message_input = torch.cat([message_input, node_states.index_select(0, dst)], -1)  # :90-92
        all_messages.append(mlp(torch.cat([message_input, feats], -1), mlp_w))     # :96-98
    aggregated = aggregate_messages(torch.cat(all_messages, 0), torch.cat(all_targets, 0),
                                    node_states.shape[0], w["agg"])               # :107-112
    x = gelu(aggregated) if w.get("gelu", True) else aggregated                   # :114-115
    if w.get("ln_w") is not None:
        x = layer_norm(x, w["ln_w"], w["ln_b"])                                   # :58
    if w.get("dense_w") is not None:
        x = linear(x, w["dense_w"], w["dense_b"])                                 # :60
        if w.get("tanh", True):
            x = torch.tanh(x)                                                     # :62-63
    return (x, aggregated) if return_aggregate else x


def layer_on_rows(node_states, adjacency_lists: Adj, w: Dict, rows: torch.Tensor):
    """`ggnn_layer` / `mlp_mp_layer` (no edge features) evaluated for the destination rows `rows` (sorted, unique
    int64) only -> [len(rows), H'].  A message-passing layer computes every output row from its own in-edges and its
    own previous state (gatedmessagepassing.py:63-69, mlpmessagepassing.py:107-117), so this is the SAME arithmetic
    -- the selected edges keep their type-major order, i.e. every row folds its messages exactly as in the whole-graph
    call -- for graphs whose [E, H] gathered message input does not fit the host (BASELINE config 5: 12.8 GB per
    shard).  tests/test_oracle_golden.py pins it to the whole-graph functions."""
    n = node_states.shape[0]
    pick = torch.zeros(n, dtype=torch.bool)
    pick[rows] = True
    targets, messages = [], []
    for t, (src, dst) in enumerate(adjacency_lists):
        sel = pick[dst]
        s, d = src[sel], dst[sel]
        targets.append(torch.searchsorted(rows, d))
        inp = node_states.index_select(0, s)
        if w["kind"] == "ggnn":
            messages.append(linear(inp, w["edge_w"][t]))                               # gatedmessagepassing.py:54-61
        else:
            if w["use_target"]:
                inp = torch.cat([inp, node_states.index_select(0, d)], -1)            # mlpmessagepassing.py:88-92
            messages.append(mlp(inp, w["edge_mlp"][t]))                                # :96-98
    agg = aggregate_messages(torch.cat(messages, 0), torch.cat(targets, 0), int(rows.shape[0]), w["agg"])
    if w["kind"] == "ggnn":
        return gru_cell(agg, node_states.index_select(0, rows), w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"])
    x = gelu(agg) if w.get("gelu", True) else agg
    if w.get("ln_w") is not None:
        x = layer_norm(x, w["ln_w"], w["ln_b"])
    if w.get("dense_w") is not None:
        x = linear(x, w["dense_w"], w["dense_b"])
        if w.get("tanh", True):
            x = torch.tanh(x)
    return x


def row_chunks(in_degree: torch.Tensor, max_edges: int):
    """Consecutive destination-row ranges [lo, hi) holding <= max_edges in-edges each (a single row beyond that gets a
    range of its own): the chunking `layer_on_rows` is driven with for a whole large graph."""
    csum = torch.cumsum(in_degree.to(torch.int64), 0)
    n, lo, base = int(in_degree.shape[0]), 0, 0
    while lo < n:
        hi = int(torch.searchsorted(csum, torch.tensor(base + max_edges), right=True))
        hi = min(n, max(hi, lo + 1))
        yield lo, hi
        base = int(csum[hi - 1])
        lo = hi


def global_gru_exchange(node_states, node_to_graph_idx, w: Dict):
    """GruGlobalStateUpdate with WeightedSum / Simple pooling (eval mode):
    globalgraphexchange.py:29-64 + varsizedsummary.py:28-41,68-81.

    w: {"pool": "weighted_sum"|"sum"|"mean"|"max"|"min", "pool_w": [1, D] (weighted_sum),
        "w_ih", "w_hh", "b_ih", "b_hh"}
    """
    num_samples = int(node_to_graph_idx.max()) + 1                                # :40
    if w["pool"] == "weighted_sum":
        weights = torch.sigmoid(linear(node_states, w["pool_w"]).squeeze(-1))     # varsizedsummary.py:73-75
        pooled = scatter(node_states * weights.unsqueeze(-1), node_to_graph_idx, dim=0,
                         dim_size=num_samples, reduce="sum")                      # :76-81
    else:
        pooled = scatter(node_states, node_to_graph_idx, dim=0, dim_size=num_samples,
                         reduce=w["pool"])                                        # :35-41
    per_node = pooled[node_to_graph_idx]                                          # globalgraphexchange.py:45
    return gru_cell(per_node, node_states, w["w_ih"], w["w_hh"], w["b_ih"], w["b_hh"])  # :62-64


# --------------------------------------------------------------------------------------------
# the container: edge augmentation + layer loop
# --------------------------------------------------------------------------------------------
def augment_adjacency(adjacency_lists: Adj, num_nodes: int, introduce_backwards_edges: bool,
                      add_self_edges: bool) -> Adj:
    """graphneuralnetwork.py:172-186.  Order: forward types, reversed types (same order), self.
    Unlike the reference this does not mutate the caller's list."""
    adj = list(adjacency_lists)
    if introduce_backwards_edges:
        adj = adj + [(t, f) for f, t in adj]                                      # :172-174
    if add_self_edges:
        idents = torch.arange(num_nodes, dtype=torch.int64)                       # :177-179
        adj.append((idents, idents))
    return adj


def run_layer_stack(node_states, adjacency_lists: Adj, layers: Sequence[Dict], edge_features=None,
                    node_to_graph_idx=None, trace: Optional[List] = None):
    """GraphNeuralNetwork.gnn, graphneuralnetwork.py:121-131 (+ residual layers,
    residuallayers.py:8-96).  `layers` is a list of specs {"kind": ..., ...}; a tied layer is
    the same dict repeated."""
    if edge_features is None:
        edge_features = [torch.empty(a[0].shape[0], 0, dtype=node_states.dtype)
                         for a in adjacency_lists]                                # :162-166
    stash: Dict[str, torch.Tensor] = {}
    x = node_states
    for spec in layers:
        kind = spec["kind"]
        x_in = x
        if kind == "ggnn":
            x = ggnn_layer(x, adjacency_lists, edge_features, spec)
        elif kind == "mlp":
            x = mlp_mp_layer(x, adjacency_lists, edge_features, spec)
        elif kind == "global_gru":
            x = global_gru_exchange(x, node_to_graph_idx, spec)
        elif kind == "residual_origin":                                           # residuallayers.py:31
            stash[spec["name"]] = x
        elif kind == "residual_concat":                                           # :86
            x = torch.cat((stash.pop(spec["name"]), x), dim=-1)
        elif kind == "residual_mean":                                             # :52
            x = torch.stack((stash.pop(spec["name"]), x), dim=-1).mean(dim=-1)
        elif kind == "residual_linear":                                           # :131
            x = linear(torch.cat((stash.pop(spec["name"]), x), dim=-1), spec["w"])
        else:
            raise ValueError(kind)
        if trace is not None:     # (layer input, layer output) per module: per-layer parity checks and
            trace.append((x_in, x))   # `return_all_states` (graphneuralnetwork.py:132-133)
    return x


def gnn_forward(initial_node_representations, adjacency_lists: Adj, layers: Sequence[Dict],
                introduce_backwards_edges: bool, add_self_edges: bool, node_to_graph_idx=None,
                trace: Optional[List] = None, edge_features=None):
    """GraphNeuralNetwork.forward minus the embedder, graphneuralnetwork.py:160-209.
    Returns (output_node_representations, num_edges_counted) where the edge count follows
    :198 (edges after augmentation)."""
    n = initial_node_representations.shape[0]
    adj = augment_adjacency(adjacency_lists, n, introduce_backwards_edges, add_self_edges)
    if edge_features is not None:   # graphneuralnetwork.py:174,183: reverse edges reuse the forward
        feats = list(edge_features)  # features, self edges get zeros of the same width
        if introduce_backwards_edges:
            feats = feats + [f for f in feats]
        if add_self_edges:
            feats.append(torch.zeros(n, feats[-1].shape[-1], dtype=feats[-1].dtype))
        edge_features = feats
    out = run_layer_stack(initial_node_representations, adj, layers, edge_features=edge_features,
                          node_to_graph_idx=node_to_graph_idx, trace=trace)
    return out, sum(int(a[0].shape[0]) for a in adj)


# --------------------------------------------------------------------------------------------
# disjoint-union batching (integer only; must be bit-exact)
# --------------------------------------------------------------------------------------------
def batch_graphs(graphs: Sequence[Dict], num_edge_types: int,
                 stop_extending_minibatch_after_num_nodes: int):
    """GraphNeuralNetworkModel.{extend_minibatch_with, finalize_minibatch},
    graphneuralnetwork.py:386-493.

    graphs: [{"num_nodes": int, "adjacency_lists": [T0 x (src int32[], dst int32[])],
              "reference_nodes": {name: int32[]}}]
    Yields minibatch dicts with int64 tensors exactly as finalize_minibatch lays them out.
    """
    def fresh():
        return {"adj": [([], []) for _ in range(num_edge_types)], "npg": [], "ref_ids": {},
                "ref_gidx": {}, "n": 0}

    def finalize(mb):
        node_to_graph = np.repeat(np.arange(len(mb["npg"]), dtype=np.int64),
                                  np.asarray(mb["npg"], dtype=np.int64))          # :440-443,469-477
        return {
            "adjacency_lists": [
                (torch.tensor(np.concatenate(f) if f else np.zeros(0, np.int32), dtype=torch.int64),
                 torch.tensor(np.concatenate(t) if t else np.zeros(0, np.int32), dtype=torch.int64))
                for f, t in mb["adj"]],                                           # :461-467
            "node_to_graph_idx": torch.tensor(node_to_graph, dtype=torch.int64),
            "reference_node_graph_idx": {k: torch.tensor(v, dtype=torch.int64)
                                         for k, v in mb["ref_gidx"].items()},     # :478-483
            "reference_node_ids": {k: torch.tensor(np.concatenate(v).astype(np.int32),
                                                   dtype=torch.int64)
                                   for k, v in mb["ref_ids"].items()},            # :484-491
            "num_graphs": len(mb["npg"]),                                         # :492
        }

    mb = fresh()
    for g in graphs:
        graph_idx = len(mb["npg"])                                                # :397
        off = mb["n"]                                                             # :401
        for (s, d), (ms, md) in zip(g["adjacency_lists"], mb["adj"]):
            ms.append(np.asarray(s, dtype=np.int32) + off)                        # :418-420
            md.append(np.asarray(d, dtype=np.int32) + off)                        # :421-423
        for name, refs in g["reference_nodes"].items():
            refs = np.asarray(refs, dtype=np.int32)
            mb["ref_gidx"].setdefault(name, []).extend(graph_idx for _ in range(len(refs)))  # :431-433
            mb["ref_ids"].setdefault(name, []).append(refs + off)                 # :434
        mb["npg"].append(int(g["num_nodes"]))                                     # :436
        mb["n"] = off + int(g["num_nodes"])                                       # :437
        if not (mb["n"] < stop_extending_minibatch_after_num_nodes):              # :438
            yield finalize(mb)
            mb = fresh()
    if mb["npg"]:
        yield finalize(mb)


# --------------------------------------------------------------------------------------------
# helpers to move layer weights around
# --------------------------------------------------------------------------------------------
def weights_from_reference_layer(layer) -> Dict:
    """Extract a weight dict from one of the *reference's own* layer objects (name-mangled
    attributes, SURVEY.md 8b 'Checkpoint compatibility')."""
    cls = type(layer).__name__
    sd = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    if cls == "GatedMessagePassingLayer":
        p = "_GatedMessagePassingLayer__"
        T = len([k for k in sd if k.startswith(p + "edge_message_transformation_layers.")])
        return {"kind": "ggnn",
                "edge_w": [sd[f"{p}edge_message_transformation_layers.{t}.weight"] for t in range(T)],
                "w_ih": sd[p + "state_update.weight_ih"], "w_hh": sd[p + "state_update.weight_hh"],
                "b_ih": sd[p + "state_update.bias_ih"], "b_hh": sd[p + "state_update.bias_hh"],
                "agg": getattr(layer, p + "aggregation_fn")}
    if cls == "MlpMessagePassingLayer":
        p = "_MlpMessagePassingLayer__"
        pref = p + "edge_message_transformation_layers."
        T = len({k[len(pref):].split(".")[0] for k in sd if k.startswith(pref)})
        edge_mlp = []
        for t in range(T):
            keys = sorted((k for k in sd if k.startswith(f"{pref}{t}._MLP__mlp_modules.")),
                          key=lambda k: int(k.split(".")[-2]))
            edge_mlp.append([sd[k] for k in keys])
        upd = getattr(layer, p + "state_update")
        spec = {"kind": "mlp", "edge_mlp": edge_mlp,
                "use_target": getattr(layer, p + "use_target_state_as_message_input"),
                "agg": getattr(layer, p + "aggregation_fn"),
                "gelu": getattr(layer, p + "message_activation") is not None,
                "ln_w": None, "ln_b": None, "dense_w": None, "dense_b": None, "tanh": False}
        for m in upd:
            n = type(m).__name__
            if n == "LayerNorm":
                spec["ln_w"], spec["ln_b"] = m.weight.detach().clone(), m.bias.detach().clone()
            elif n == "Linear":
                spec["dense_w"], spec["dense_b"] = m.weight.detach().clone(), m.bias.detach().clone()
            elif n == "Tanh":
                spec["tanh"] = True
        return spec
    if cls == "GruGlobalStateUpdate":
        p = "_GruGlobalStateUpdate__gru_cell."
        pool = getattr(layer, "_AbstractGlobalGraphExchange__global_graph_representation_module")
        spec = {"kind": "global_gru", "w_ih": sd[p + "weight_ih"], "w_hh": sd[p + "weight_hh"],
                "b_ih": sd[p + "bias_ih"], "b_hh": sd[p + "bias_hh"]}
        if type(pool).__name__ == "WeightedSumVarSizedElementReduce":
            spec["pool"] = "weighted_sum"
            spec["pool_w"] = pool.state_dict()["_WeightedSumVarSizedElementReduce__weights_layer.weight"].clone()
        else:
            spec["pool"] = getattr(pool, "_SimpleVarSizedElementReduce__summarization_type")
        return spec
    raise TypeError(cls)


def cast_spec(spec: Dict, dtype) -> Dict:
    """Deep-cast every floating tensor in a layer spec (used for the fp64 error attribution)."""
    def c(v):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            return v.to(dtype)
        if isinstance(v, (list, tuple)):
            return type(v)(c(u) for u in v)
        return v
    return {k: c(v) for k, v in spec.items()}
