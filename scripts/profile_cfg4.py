"""configs[3] (VarMisuse MLP-MP stack, hidden 64, T = 21) on ONE GPU, unsharded: the loop rocprofv3 wraps to see
where a cfg4 forward goes (bench.py times the same stack through sharded.run_stack).
usage: python scripts/profile_cfg4.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ptgnn_amd import layers as L, ops, workloads
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    H, T = 64, 21
    mb = workloads.batched_graphs(40, 2000, 10, 2.4, seed=21)
    n = mb["num_nodes"]
    adj = list(mb["adjacency_lists"])
    adj = adj + [(d_, s_) for s_, d_ in adj]
    ar = torch.arange(n, dtype=torch.int64)
    adj.append((ar, ar))
    adj = [(s.to(dev), d.to(dev)) for s, d in adj]
    n2g = mb["node_to_graph_idx"].to(dev)
    torch.manual_seed(4)
    mk = lambda: L.MlpMessagePassingLayer(H, H, H, T, "max", dropout_rate=0.1)          # noqa: E731
    mk2 = lambda: L.MlpMessagePassingLayer(2 * H, H, 2 * H, T, "max", dropout_rate=0.1)  # noqa: E731
    r1, r2, r3 = L.ConcatResidualLayer(H), L.MeanResidualLayer(H), L.ConcatResidualLayer(H)
    mods = [r1.pass_through_dummy_layer(), mk(), mk(), mk(), r1, mk2(), r2.pass_through_dummy_layer(), mk(), mk(), r2,
            r3.pass_through_dummy_layer(), mk(), r3, mk2()]
    mods = [m.to(dev).eval() for m in mods]
    x0 = workloads.node_states(n, H, seed=6).to(dev)
    feats = [None] * len(adj)

    def step():
        ops.clear_plan_cache()
        x = x0
        with torch.no_grad(), L.forward_scope():
            for m in mods:
                x = m(x, adj, n2g, {}, {}, feats)
        return x

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    e = sum(int(a[0].shape[0]) for a in adj)
    print(f"cfg4 single GPU: N={n} E={e} T={T}: {dt * 1e3:.3f} ms per forward ({dt * 1e3 / 8:.3f} ms per MP layer)")


if __name__ == "__main__":
    main()
