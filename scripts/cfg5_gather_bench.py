"""cfg5 per-GPU shard (power-law N=1.25M, E=12.5M, M=256): the aggregation launch on its own (HIP events, median).
PTGNN_AMD_HUB_STREAM=0 keeps the hub / long-row launches on the caller's stream (round-2 behaviour) for A/B."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops, workloads  # noqa: E402

N, E, H = 1_250_000, 12_500_000, 256
adj = workloads.power_law_graph(N, E, alpha=0.8, seed=1234)
cadj = [(adj[0][0].cuda(), adj[0][1].cuda())]
plan = ops.build_plan(cadj, N)
y = torch.randn(N, H, device="cuda")
res = {"hub_stream": os.environ.get("PTGNN_AMD_HUB_STREAM", "1")}
nbytes = E * (4.0 * H + 4) + N * (4.0 * H + 4)
for red in ("sum", "max"):
    for _ in range(3):
        ops.gather_reduce(y, plan, H, red)
    evs = []
    for _ in range(11):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.gather_reduce(y, plan, H, red); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[5]
    res[red] = {"ms": round(ms, 4), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000.0, 4)}
print(json.dumps(res))
