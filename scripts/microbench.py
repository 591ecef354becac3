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
