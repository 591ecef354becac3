"""Hidden-64 shapes of the README's default architecture (MLP-MP, hidden 64) and of BASELINE config 4, one op at a time:
HIP events, [min, median] over interleaved repetitions after a burn.  Run ON THE GPU BOX.  PTGNN_AMD_LIB selects a build."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptgnn_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(3)
N = 115772
dev = "cuda"


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev)


x64, x128, g64, g128 = rnd(N, 64), rnd(N, 128), rnd(N, 64), rnd(N, 128)
w6464, w64128, w12864, b64 = rnd(64, 64) / 8, rnd(64, 128) / 11, rnd(128, 64) / 8, rnd(64)
gam64, bet64, gam128, bet128 = rnd(64), rnd(64), rnd(128), rnd(128)
EPI = 3   # GELU + LayerNorm


def forced(flag, fn):
    def run():
        os.environ["PTGNN_AMD_FORCE_STREAM"] = flag
        fn()
    return run


from ptgnn_amd import workloads  # noqa: E402
mb = workloads.batched_graphs(48, 2500, 8, 2.2, seed=1234)
adj = [(s_.cuda(), d_.cuda()) for s_, d_ in mb["adjacency_lists"]]
adj = adj + [(d_, s_) for s_, d_ in adj]
ar = torch.arange(N, device="cuda")
adj.append((ar, ar))
plan = ops.plan_for(adj, N)
E = plan.num_edges
m64, m128 = rnd(E, 64), rnd(E, 128)
_, arg64 = ops.segment_reduce(m64, plan, "max", return_arg=True)
_, arg128 = ops.segment_reduce(m128, plan, "max", return_arg=True)

cases = {
    "spread_max_128": lambda: ops.segment_spread(g128, arg128, plan),
    "spread_sum_128": lambda: ops.segment_spread(g128, None, plan),
    "spread_max_64": lambda: ops.segment_spread(g64, arg64, plan),
    "segreduce_max_128": lambda: ops.segment_reduce(m128, plan, "max"),
    "segreduce_max_64": lambda: ops.segment_reduce(m64, plan, "max"),
    "copy_E128": lambda: m128.copy_(m128),
    "linear_64_64_tanh_tile": forced("0", lambda: ops.linear(x64, w6464, b64, act="tanh")),
    "linear_64_64_tanh_stream": forced("1", lambda: ops.linear(x64, w6464, b64, act="tanh")),
    "linear_64_64_tile": forced("0", lambda: ops.linear(g64, w6464)),
    "linear_64_64_stream": forced("1", lambda: ops.linear(g64, w6464)),
    "linear_128_64_tile": forced("0", lambda: ops.linear(x128, w64128)),
    "linear_128_64_stream": forced("1", lambda: ops.linear(x128, w64128)),
    "linear_64_128_tile": forced("0", lambda: ops.linear(x64, w12864)),
    "linear_64_128_stream": forced("1", lambda: ops.linear(x64, w12864)),
    "wgrad_64x64_bias": lambda: ops.linear_weight_grad(x64, g64, want_bias=True),
    "wgrad_128x64": lambda: ops.linear_weight_grad(x128, g64),
    "wgrad_128x128_bias": lambda: ops.linear_weight_grad(x128, g128, want_bias=True),
    "row_epi_64": lambda: ops.row_epilogue(x64, EPI, gam64, bet64, 1e-5),
    "row_epi_128": lambda: ops.row_epilogue(x128, EPI, gam128, bet128, 1e-5),
    "row_epi_bwd_64": lambda: ops.row_epilogue_backward(x64, g64, EPI, gam64, 1e-5),
    "row_epi_bwd_128": lambda: ops.row_epilogue_backward(x128, g128, EPI, gam128, 1e-5),
    "copy_64": lambda: g64.copy_(x64),
}
for _ in range(40):
    for fn in cases.values():
        fn()
torch.cuda.synchronize()
reps = int(os.environ.get("PROBE_REPS", "31"))
evs = []
for _ in range(reps):
    for k, fn in cases.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((k, s, e))
torch.cuda.synchronize()
times = {k: [] for k in cases}
for k, s, e in evs:
    times[k].append(s.elapsed_time(e) * 1e3)
res = {"lib": os.path.basename(os.environ.get("PTGNN_AMD_LIB", "default"))}
for k, v in times.items():
    v.sort()
    res[k] = [round(v[0], 1), round(v[len(v) // 2], 1)]
print(json.dumps(res))
