"""Time the shared-row / per-edge edge GEMM and the fused GRU on the cfg3 batch: HIP events, min and median over
interleaved repetitions after a clock-ramping burn.  PTGNN_AMD_LIB selects a probe build (scripts/build_variant.sh <tag>
stream_gemm.hip -D...).  Run ON THE GPU BOX.
(The `*_v2` / `*_old` pairs alternated the round-4 fence-free kernel with k_stream_edge through the PTGNN_AMD_EDGE_V2 knob
of the probe builds of profiles/r04_notes.md section 1; the shipped library ignores the knob -- k_stream_edge_v2 is only
instantiated for the dropout forms -- so both names time k_stream_edge there.)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptgnn_amd import ops, workloads  # noqa: E402

mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
N = mb["num_nodes"]
adj = [(s.cuda(), d.cuda()) for s, d in mb["adjacency_lists"]]
adj = adj + [(d, s) for s, d in adj]
ar = torch.arange(N, device="cuda")
adj.append((ar, ar))
g = torch.Generator().manual_seed(3)
K = int(os.environ.get("PROBE_K", "128"))
x = torch.randn(N, K, generator=g).cuda()
ws = [(torch.randn(128, K, generator=g) / 11.3).cuda() for _ in adj]
plan = ops.plan_for(adj, N)
uq = plan.unique_messages()
cell = torch.nn.GRUCell(128, 128).cuda()
a = torch.randn(N, 128, generator=g).cuda()
h = torch.randn(N, 128, generator=g).cuda()


def with_v2(flag, fn):
    def run():
        os.environ["PTGNN_AMD_EDGE_V2"] = flag
        fn()
    return run


cases = {
    "shared_v2": with_v2("1", lambda: ops.edge_linear_shared(x, uq, ws)),
    "shared_old": with_v2("0", lambda: ops.edge_linear_shared(x, uq, ws)),
    "per_edge_v2": with_v2("1", lambda: ops.edge_linear(x, adj, ws, False)),
    "per_edge_old": with_v2("0", lambda: ops.edge_linear(x, adj, ws, False)),
    "gru": lambda: ops.gru_cell(a, h, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh),
}
for _ in range(60):                      # ~0.1 s of back-to-back GEMMs: clocks and caches in steady state
    for fn in cases.values():
        fn()
torch.cuda.synchronize()
reps = int(os.environ.get("PROBE_REPS", "41"))
times = {k: [] for k in cases}
evs = []
for _ in range(reps):
    for k, fn in cases.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((k, s, e))
torch.cuda.synchronize()
for k, s, e in evs:
    times[k].append(s.elapsed_time(e) * 1e3)
res = {"lib": os.path.basename(os.environ.get("PTGNN_AMD_LIB", "default")), "K": K, "rows": uq.rows(wait=True)}
for k, v in times.items():
    v.sort()
    res[k] = [round(v[0], 1), round(v[len(v) // 2], 1)]
print(json.dumps(res))
