#!/usr/bin/env python
"""Correctness + speed of the dense blocks per GEMM mode (0 tile, 1 stream fp32, 2 stream 3xbf16 split) on the
BASELINE shapes.  Run ON THE GPU BOX:  python scripts/gemm_bench.py [out.json]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ptgnn_amd import ops, workloads  # noqa: E402


def clock(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return min(ts), sorted(ts)[1]


VARIANTS = [("m0", 0), ("m1", 1), ("m2", 2)]


def variants():
    for name, mode in VARIANTS:
        ops.set_gemm_mode(mode)
        yield name


def main():
    out = {"linear": [], "gru": [], "edge": []}
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    # ---- linear
    for rows, k, n_out, act, bias in [(200_000, 128, 256, None, False), (200_000, 128, 128, "tanh", True),
                                      (115_772, 128, 128, None, False), (115_772, 256, 128, None, False),
                                      (80_001, 64, 64, "relu", True), (5_003, 192, 96, None, True),
                                      (115_772, 128, 2176, None, False), (33, 64, 32, None, False)]:
        x = torch.randn(rows, k, generator=g).to(dev)
        w = (torch.randn(n_out, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n_out, generator=g).to(dev) if bias else None
        ref = x.double() @ w.double().t()
        if b is not None:
            ref = ref + b.double()
        if act == "tanh":
            ref = torch.tanh(ref)
        elif act == "relu":
            ref = torch.relu(ref)
        row = {"shape": [rows, k, n_out], "act": act}
        for name in variants():
            y = ops.linear(x, w, b, act=act)
            err = float((y.double() - ref).abs().max())
            tmin, tmed = clock(lambda: ops.linear(x, w, b, act=act))
            row[name] = {"err64": err, "us": round(tmin * 1e6, 1), "us_med": round(tmed * 1e6, 1),
                               "tflops": round(2.0 * rows * k * n_out / tmin / 1e12, 1)}
        out["linear"].append(row)
        print(row, flush=True)
        del x, w, ref
    # ---- GRU
    for n, m, h in [(115_772, 128, 128), (115_772, 128, 256), (80_003, 64, 64), (200_000, 128, 128), (77, 64, 64)]:
        a = torch.randn(n, m, generator=g).to(dev)
        hh = torch.randn(n, h, generator=g).to(dev)
        cell = torch.nn.GRUCell(m, h).to(dev)
        with torch.no_grad():
            ref = torch.nn.GRUCell(m, h).double().to(dev)
            ref.load_state_dict({k_: v.double() for k_, v in cell.state_dict().items()})
            want = ref(a.double(), hh.double())
        row = {"shape": [n, m, h]}
        for name in variants():
            f = lambda: ops.gru_cell(a, hh, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)  # noqa: E731
            err = float((f().double() - want).abs().max())
            tmin, tmed = clock(f)
            row[name] = {"err64": err, "us": round(tmin * 1e6, 1), "us_med": round(tmed * 1e6, 1),
                               "tflops": round(2.0 * n * 3 * h * (m + h) / tmin / 1e12, 1)}
        out["gru"].append(row)
        print(row, flush=True)
    # ---- grouped edge GEMM on the Graph2Class batch
    mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
    N = mb["num_nodes"]
    adj = [(s.to(dev), d.to(dev)) for s, d in mb["adjacency_lists"]]
    adj = adj + [(d, s) for s, d in adj]
    ar = torch.arange(N, device=dev)
    adj.append((ar, ar))
    E = sum(int(s.shape[0]) for s, _ in adj)
    for H, M, use_dst in [(128, 128, False), (256, 128, False), (64, 64, True), (128, 128, True)]:
        x = torch.randn(N, H, generator=g).to(dev)
        K = H * (2 if use_dst else 1)
        ws = [(torch.randn(M, K, generator=g) / K ** 0.5).to(dev) for _ in adj]
        parts = []
        for (s, d), w in zip(adj, ws):
            inp = x[s].double()
            if use_dst:
                inp = torch.cat([inp, x[d].double()], 1)
            parts.append(inp @ w.double().t())
        ref = torch.cat(parts)
        row = {"shape": [E, K, M], "use_dst": use_dst, "T": len(adj)}
        for name in variants():
            f = lambda: ops.edge_linear(x, adj, ws, use_dst)  # noqa: E731
            err = float((f().double() - ref).abs().max())
            tmin, tmed = clock(f)
            row[name] = {"err64": err, "us": round(tmin * 1e6, 1), "us_med": round(tmed * 1e6, 1),
                               "tflops": round(2.0 * E * K * M / tmin / 1e12, 1)}
        out["edge"].append(row)
        print(row, flush=True)
        del ref, parts
    ops.set_gemm_mode(1)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
