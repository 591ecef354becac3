"""Host-side overhead check (run on the GPU box): wall vs GPU time of one cfg3 forward + cProfile."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchmarks import graph2class as bench  # noqa: E402

dev = torch.device("cuda:0")
st = bench.make_cfg3(dev)
for _ in range(3):
    bench.step_cfg3(st)
torch.cuda.synchronize()
for name in ("wall_sync_each", "wall_async"):
    t0 = time.perf_counter()
    for _ in range(10):
        bench.step_cfg3(st)
        if name == "wall_sync_each":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms/forward")
t0 = time.perf_counter()
for _ in range(10):
    bench.step_cfg3(st)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue time", (t1 - t0) / 10 * 1e3, "ms/forward; total", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.step_cfg3(st)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
