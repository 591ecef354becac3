"""Kernel micro-benchmarks on the GPU box (HIP events, median of N): python scripts/microbench.py"""
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops  # noqa: E402


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts)


def linear_shapes():
    shapes = [(200_000, 128, 256), (200_000, 128, 128), (115_772, 128, 2176), (115_772, 256, 2176),
              (200_000, 64, 128), (200_000, 256, 256), (8192, 4096, 4096)]
    if os.environ.get("PTGNN_AMD_ABLATE"):
        shapes = shapes[:3]
    for rows, k, n in shapes:
        x = torch.randn(rows, k, device="cuda"); w = torch.randn(n, k, device="cuda")
        out = torch.empty(rows, n, device="cuda")
        ms = timeit(lambda: ops.linear(x, w, out=out))
        print(f"linear rows={rows} k={k} n={n}: {ms*1e3:.1f} us  {2.0*rows*k*n/ms/1e9:.1f} TFLOP/s  ablate={os.environ.get('PTGNN_AMD_ABLATE')}")
        if os.environ.get("PTGNN_AMD_ABLATE"):
            continue
        t = timeit(lambda: torch.mm(x, w.t(), out=out))
        print(f"   torch.mm (rocBLAS/hipBLASLt) same shape: {t*1e3:.1f} us  {2.0*rows*k*n/t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "linear"
    if what == "linear":
        linear_shapes()


def powerlaw():
    """BASELINE config 5 at its per-GPU size (8-way dst-range shard of N=10M / E=100M, H=256):
    1.25M rows, 12.5M in-edges with power-law destinations."""
    from ptgnn_amd import workloads
    N, E, M = 1_250_000, 12_500_000, 256
    for alpha in (0.8, 1.1):
        adj = workloads.power_law_graph(N, E, alpha=alpha, seed=1)
        deg = torch.bincount(adj[0][1], minlength=N)
        cadj = [(adj[0][0].cuda(), adj[0][1].cuda())]
        y = torch.randn(N, M, device="cuda")
        for thr in (4096, 0):
            ops.HUB_THRESHOLD = thr
            plan = ops.build_plan(cadj, N)
            for red in ("sum", "max"):
                ms = timeit(lambda: ops.gather_reduce(y, plan, M, red), n=5, warm=2)
                nbytes = E * (4.0 * M + 4) + N * (4.0 * M + 4)
                print(f"powerlaw alpha={alpha} max_deg={int(deg.max())} hubs>{4096}={int((deg > 4096).sum())} "
                      f"hub_threshold={thr} {red}: {ms:.3f} ms  {E / ms / 1e6:.2f} G edges/s  {nbytes / ms / 1e9:.2f} TB/s")
            t = timeit(lambda: ops.build_plan(cadj, N), n=5, warm=2)
            print(f"   plan build: {t:.3f} ms")
        del y, plan


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "powerlaw":
    powerlaw()
