"""Test helpers: build ptgnn_amd layers from oracle layer specs (golden fixtures / random)."""
import torch
from torch import nn

from ptgnn_amd import layers as L


def layer_from_spec(spec):
    if spec["kind"] == "ggnn":
        T = len(spec["edge_w"])
        M, H = spec["edge_w"][0].shape
        lay = L.GatedMessagePassingLayer(H, M, T, spec["agg"])
        sd = lay.state_dict()
        p = "_GatedMessagePassingLayer__"
        for t in range(T):
            sd[f"{p}edge_message_transformation_layers.{t}.weight"] = spec["edge_w"][t]
        sd[p + "state_update.weight_ih"], sd[p + "state_update.weight_hh"] = spec["w_ih"], spec["w_hh"]
        sd[p + "state_update.bias_ih"], sd[p + "state_update.bias_hh"] = spec["b_ih"], spec["b_hh"]
        lay.load_state_dict(sd)
        return lay
    if spec["kind"] == "mlp":
        T = len(spec["edge_mlp"])
        ws = spec["edge_mlp"][0]
        M = ws[-1].shape[0]
        in_w = ws[0].shape[1]
        H = in_w // 2 if spec["use_target"] else in_w
        hidden = [w.shape[0] for w in ws[:-1]]
        out_dim = spec["dense_w"].shape[0] if spec["dense_w"] is not None else M
        lay = L.MlpMessagePassingLayer(
            H, out_dim, M, T, spec["agg"],
            message_activation=nn.GELU() if spec["gelu"] else None,
            use_target_state_as_message_input=spec["use_target"], mlp_hidden_layers=hidden,
            use_layer_norm=spec["ln_w"] is not None, use_dense_layer=spec["dense_w"] is not None,
            dense_activation=nn.Tanh() if spec["tanh"] else None)
        sd = lay.state_dict()
        p = "_MlpMessagePassingLayer__"
        for t in range(T):
            keys = sorted((k for k in sd if k.startswith(f"{p}edge_message_transformation_layers.{t}.")),
                          key=lambda k: int(k.split(".")[-2]))
            for k, w in zip(keys, spec["edge_mlp"][t]):
                sd[k] = w
        idx = 0
        if spec["ln_w"] is not None:
            sd[f"{p}state_update.{idx}.weight"], sd[f"{p}state_update.{idx}.bias"] = spec["ln_w"], spec["ln_b"]
            idx += 1
        if spec["dense_w"] is not None:
            sd[f"{p}state_update.{idx}.weight"], sd[f"{p}state_update.{idx}.bias"] = spec["dense_w"], spec["dense_b"]
        lay.load_state_dict(sd)
        return lay
    raise ValueError(spec["kind"])


def stack_from_specs(specs):
    """Oracle spec list (with tied layers and residual markers) -> list of ptgnn_amd modules."""
    built, residuals, mods = {}, {}, []
    for i, spec in enumerate(specs):
        k = spec["kind"]
        if k in ("ggnn", "mlp"):
            if id(spec) not in built:
                built[id(spec)] = layer_from_spec(spec)
            mods.append(built[id(spec)])
        elif k == "global_gru":
            from ptgnn_amd import reduceops as R
            H = spec["w_hh"].shape[1]
            D = spec["w_ih"].shape[1]
            pool = (R.WeightedSumVarSizedElementReduce(H) if spec["pool"] == "weighted_sum"
                    else R.SimpleVarSizedElementReduce(spec["pool"]))
            lay = R.GruGlobalStateUpdate(pool, H, D)
            sd = lay.state_dict()
            p = "_GruGlobalStateUpdate__gru_cell."
            sd[p + "weight_ih"], sd[p + "weight_hh"] = spec["w_ih"], spec["w_hh"]
            sd[p + "bias_ih"], sd[p + "bias_hh"] = spec["b_ih"], spec["b_hh"]
            if spec["pool"] == "weighted_sum":
                key = [k2 for k2 in sd if k2.endswith("weights_layer.weight")][0]
                sd[key] = spec["pool_w"]
            lay.load_state_dict(sd)
            mods.append(lay)
        elif k == "residual_origin":
            mods.append(("origin", spec["name"], len(mods)))
        elif k in ("residual_concat", "residual_mean"):
            # find the dim of the stashed state: output dim of whatever precedes the origin marker
            origin_pos = next(m[2] for m in mods if isinstance(m, tuple) and m[1] == spec["name"])
            dim = _dim_before(mods, origin_pos, specs)
            res = (L.ConcatResidualLayer if k == "residual_concat" else L.MeanResidualLayer)(dim)
            mods[origin_pos] = res.pass_through_dummy_layer()
            mods.append(res)
        else:
            raise ValueError(k)
    return mods


def _dim_before(mods, pos, specs):
    for m in reversed(mods[:pos]):
        if not isinstance(m, tuple):
            return m.output_state_dimension
    for m in mods[pos + 1:]:
        if not isinstance(m, tuple):
            return m.input_state_dimension
    raise ValueError


def to_cuda_adj(adj):
    return [(s.cuda(), d.cuda()) for s, d in adj]


def empty_feats(adj, device):
    return [torch.empty(a[0].shape[0], 0, device=device) for a in adj]


def dropout_keep_scale(seed: int, num_rows: int, width: int, p: float) -> torch.Tensor:
    """numpy restatement of ptgnn_amd/csrc/dense_common.h `dropout_apply4`: the [num_rows, width] fp32
    multiplier (0 or 1/(1-p)) the HIP kernels apply to the gathered GGNN message input."""
    import numpy as np

    def mix32(v):
        v = v.astype(np.uint32)
        v ^= v >> np.uint32(16)
        v = (v * np.uint32(0x21F0AAAD)).astype(np.uint32)
        v ^= v >> np.uint32(15)
        v = (v * np.uint32(0x735A2D97)).astype(np.uint32)
        v ^= v >> np.uint32(15)
        return v

    assert width % 2 == 0
    half = width // 2
    thr = int(p * 65536.0 + 0.5)
    with np.errstate(over="ignore"):
        idx = (np.arange(num_rows, dtype=np.uint64)[:, None] * np.uint64(half)
               + np.arange(half, dtype=np.uint64)[None, :])
        lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        h = mix32(lo ^ np.uint32(seed & 0xFFFFFFFF))
        h = mix32((h + hi * np.uint32(0x9E3779B9) + np.uint32((seed >> 32) & 0xFFFFFFFF)).astype(np.uint32))
    keep = np.empty((num_rows, width), dtype=bool)
    keep[:, 0::2] = (h & np.uint32(0xFFFF)) >= thr
    keep[:, 1::2] = (h >> np.uint32(16)) >= thr
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(keep.astype(np.float32) * scale)


def golden_batcher_graphs(g):
    """The tensorized graphs of tests/golden/batcher.npz as (adjacency_lists, num_nodes, reference_nodes)."""
    T0 = len(g["edge_type_order"])
    graphs = []
    for gi in range(int(g["num_graphs_in"])):
        refs = {k.split(".")[-1]: g[k] for k in g.files if k.startswith(f"g{gi}.ref.")}
        graphs.append(([(g[f"g{gi}.adj.{t}.src"], g[f"g{gi}.adj.{t}.dst"]) for t in range(T0)],
                       int(g[f"g{gi}.num_nodes"]), refs))
    return T0, graphs


def replay_minibatches(builder_cls, T0, graphs, stop_after, finalize):
    """The reference's minibatch loop (abstractneuralmodel.py:290-319) over a MinibatchBuilder."""
    out, b = [], builder_cls(T0, stop_after)
    for adj, n, refs in graphs:
        if not b.extend(adj, n, refs):
            out.append(finalize(b))
            b = builder_cls(T0, stop_after)
    if len(b):
        out.append(finalize(b))
    return out
