"""BASELINE configs[3]: the VarMisuse batch (40 graphs x ~2000 nodes, T0 = 10 -> T = 21) on ONE GPU through both stacks of
ptgnn/implementations/varmisuse/train.py: the shipped 8-layer MLP-MP stack (:42-74) and the GGNN variant with
GruGlobalStateUpdate (:76-107)."""
import time

import torch

from benchmarks.common import PARITY_TOL, _log


def cfg4_batch():
    """configs[3]: the VarMisuse batch (40 graphs x ~2000 nodes, T0 = 10) with reverse and self edges: T = 21."""
    from ptgnn_amd import workloads
    mb = workloads.batched_graphs(40, 2000, 10, 2.4, seed=21)
    n, n2g = mb["num_nodes"], mb["node_to_graph_idx"]
    adj = list(mb["adjacency_lists"])
    adj = adj + [(d_, s_) for s_, d_ in adj]
    ar = torch.arange(n, dtype=torch.int64)
    adj.append((ar, ar))
    return mb, adj, n, n2g


def cfg4_modules(dev, H=64, T=21, with_specs=False):
    """The shipped stack of create_var_misuse_gnn_model (varmisuse/train.py:42-74): [origin r1, MLP x3, r1 (concat),
    MLP(2H -> H, M = 2H), origin r2, MLP x3, r2 (concat), MLP(2H -> H)] -- 8 MLP-MP layers in 12 modules, hidden 64, max,
    dropout 0.1 (eval).  `with_specs`: also the oracle's layer-spec list of the same stack (oracle/mp_oracle.py
    run_layer_stack).  (Rounds 2-5 timed a 14-module variant with a mean residual in the middle -- the same eight layers;
    round 6 follows the factory module for module.)"""
    from ptgnn_amd import layers as L
    torch.manual_seed(4)
    mk = lambda: L.MlpMessagePassingLayer(H, H, H, T, "max", dropout_rate=0.1)          # noqa: E731
    mk2 = lambda: L.MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max", dropout_rate=0.1)  # noqa: E731
    r1, r2 = L.ConcatResidualLayer(H), L.ConcatResidualLayer(H)
    mods = [r1.pass_through_dummy_layer(), mk(), mk(), mk(), r1, mk2(),
            r2.pass_through_dummy_layer(), mk(), mk(), mk(), r2, mk2()]
    marks = [("residual_origin", "r1"), None, None, None, ("residual_concat", "r1"), None,
             ("residual_origin", "r2"), None, None, None, ("residual_concat", "r2"), None]
    specs = [m.export_weights() if mk_ is None else {"kind": mk_[0], "name": mk_[1]} for m, mk_ in zip(mods, marks)]
    mods = [m.to(dev).eval() for m in mods]
    return (mods, specs) if with_specs else mods


def cfg4_ggnn_modules(dev, H=64, T=21, with_specs=False):
    """The GGNN variant of the same factory (varmisuse/train.py:76-107): ONE tied GatedMessagePassingLayer (sum, dropout
    0.01) applied eight times, two GruGlobalStateUpdate layers over WeightedSumVarSizedElementReduce pools (dropout 0.1)
    and two mean residuals that both start at the input:
    [origin r1, origin r2, ggnn x3, global, ggnn, r1 (mean), ggnn x3, global, ggnn, r2 (mean)]."""
    from ptgnn_amd import layers as L, reduceops as R
    torch.manual_seed(14)
    ggnn = L.GatedMessagePassingLayer(H, H, T, "sum", dropout_rate=0.01)
    r1, r2 = L.MeanResidualLayer(H), L.MeanResidualLayer(H)
    glob = lambda: R.GruGlobalStateUpdate(R.WeightedSumVarSizedElementReduce(H), H, H, dropout_rate=0.1)  # noqa: E731
    g1, g2 = glob(), glob()
    mods = [r1.pass_through_dummy_layer(), r2.pass_through_dummy_layer(), ggnn, ggnn, ggnn, g1, ggnn, r1,
            ggnn, ggnn, ggnn, g2, ggnn, r2]
    marks = [("residual_origin", "r1"), ("residual_origin", "r2"), None, None, None, None, None, ("residual_mean", "r1"),
             None, None, None, None, None, ("residual_mean", "r2")]
    tied = ggnn.export_weights()
    specs = [(tied if m is ggnn else m.export_weights()) if mk_ is None else {"kind": mk_[0], "name": mk_[1]}
             for m, mk_ in zip(mods, marks)]
    mods = [m.to(dev).eval() for m in mods]
    return (mods, specs) if with_specs else mods


def config4(dev, k=20, parity=True, arch="mlp"):
    """configs[3] on ONE GPU, unsharded: a VarMisuse stack over the whole batch through the container's layer loop
    (the 4-GPU dst-range-sharded form is `cut_edges_variant.cfg4_stack` at N > 1).  arch "mlp": the shipped 8-layer MLP-MP
    stack; "ggnn": the GGNN variant with global exchange.  Parity, at the full benchmarked size: every message-passing /
    global-exchange layer fed the ORACLE's input of that layer (the stated 1e-5 bar), and the whole stack attributed
    against a float64 evaluation (8 stacked LayerNorms amplify fp32 rounding: the reference's own fp32 arithmetic sits
    ~7e-5 from float64 end to end on the MLP stack, so "within 1e-5 of the reference" is not a property any fp32
    implementation of that stack can have; the HIP path must be no further from float64 than 2 x the oracle is)."""
    from benchmarks.common import attributed_parity
    from ptgnn_amd import ops, workloads
    mb, adj_cpu, n, n2g_cpu = cfg4_batch()
    adj = [(s_.to(dev), d_.to(dev)) for s_, d_ in adj_cpu]
    n2g = n2g_cpu.to(dev)
    mods, specs = (cfg4_modules if arch == "mlp" else cfg4_ggnn_modules)(dev, with_specs=True)
    x_cpu = workloads.node_states(n, 64, seed=6)
    x0 = x_cpu.to(dev)
    feats = [None] * len(adj)
    # the layer loop of the container (graphneuralnetwork.py:122-131), as ptgnn runs a stack
    from ptgnn_amd.gnn import GraphNeuralNetwork
    net = GraphNeuralNetwork(mods, torch.nn.Identity(), False, False).to(dev).eval()

    def step():
        ops.clear_plan_cache()
        with torch.no_grad():
            return net.gnn(x0, adj, feats, n2g, {}, {})
    for _ in range(3):
        step()
    blocks = []          # median of three k-step blocks: one host hiccup inside a single block moved this figure from 1.19 to
    for _ in range(3):   # 2.13 ms between two runs of the same tree (round 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / k)
    dt = sorted(blocks)[1]
    e = sum(int(a[0].shape[0]) for a in adj)
    what = ("8 MLP-MP layers hidden 64 (+ 2 concat residuals), max" if arch == "mlp" else
            "8 GGNN layers (one tied instance) hidden 64, sum, + 2 GruGlobalStateUpdate over weighted-sum pools + 2 mean "
            "residuals")
    res = {"workload": f"cfg4: VarMisuse batch N={n}, T=21, E={e} (incl. reverse+self), {what}, one GPU, unsharded",
           "ms_per_forward": round(dt * 1e3, 4), "ms_per_forward_is": f"median of 3 blocks of {k} forwards",
           "ms_per_forward_blocks": [round(b * 1e3, 4) for b in blocks], "edges_per_sec_per_layer": round(e / (dt / 8), 1),
           "nodes_per_sec_per_layer": round(n / (dt / 8), 1), "edges_per_sec_readme_convention": round(e / dt, 1)}
    if parity:
        from oracle import mp_oracle as O
        with torch.no_grad():
            trace = []
            want = O.run_layer_stack(x_cpu, adj_cpu, specs, node_to_graph_idx=n2g_cpu, trace=trace)
            exact = O.run_layer_stack(x_cpu.double(), adj_cpu, [O.cast_spec(sp, torch.float64) for sp in specs],
                                      node_to_graph_idx=n2g_cpu)
            worst, pools = 0.0, []
            for mod, spec, (x_in, x_out) in zip(mods, specs, trace):
                if spec["kind"] not in ("mlp", "ggnn", "global_gru"):
                    continue
                ops.clear_plan_cache()
                got = mod(x_in.to(dev), adj, n2g, {}, {}, feats).cpu()
                if spec["kind"] == "global_gru":
                    # a pool adds ~2 000 fp32 node states per graph: ANY two fold orders differ by ~3e-5 behind the GRU
                    # (the oracle's own serial order sits that far from float64), so this layer is attributed
                    ex = O.global_gru_exchange(x_in.double(), n2g_cpu, O.cast_spec(spec, torch.float64))
                    pools.append(attributed_parity(got, x_out, ex))
                else:
                    worst = max(worst, float((got - x_out).abs().max()))
            got = step().cpu()
        rec = attributed_parity(got, want, exact)
        rec.update(per_layer_max=worst, end_to_end=rec["max_abs"], n=n,
                   ok=bool(worst <= PARITY_TOL and rec["ok"] and all(p_["ok"] for p_ in pools)),
                   against="oracle/mp_oracle.py at full size: per layer (each layer fed the oracle's input of that "
                           "layer) and end to end, attributed against a float64 evaluation of the stack")
        if pools:
            rec["global_exchange_layers"] = {"layers": len(pools), "max_abs": max(p_["max_abs"] for p_ in pools),
                                             "ours_vs_fp64": max(p_["ours_vs_fp64"] for p_ in pools),
                                             "oracle_vs_fp64": max(p_["oracle_vs_fp64"] for p_ in pools),
                                             "ok": all(p_["ok"] for p_ in pools)}
        res["parity"] = rec
    return res
